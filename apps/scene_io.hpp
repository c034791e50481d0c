// scene_io.hpp -- host-side file formats around the hot path: a Wavefront .obj reader that
// yields the flat triangle / colour / emission arrays IntersectorOctreeGPU::build consumes (the role of
// voxUtil.hpp:19-61 trianglesFlattened over prlib's readers, which are absent), and PPM / uncompressed-PNG writers
// (the role of pr::Image2DRGBA8::saveAsPngUncompressed, RTCamp.cpp:189-191).  Plain C++17, no dependencies.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mvrt_io
{
struct V3
{
	float x, y, z;
};

// `v x y z [r g b]`, `f a b c ...` with a, a/t, a//n, a/t/n and negative (relative) indices; polygons are fanned.
// Colours default to white, emissions to black (voxUtil.hpp:49-61).  Returns false if the file cannot be read.
inline bool readObj( const char* path, std::vector<V3>* vertices, std::vector<V3>* vcolors, std::vector<V3>* vemissions )
{
	FILE* fp = std::fopen( path, "rb" );
	if( !fp ) return false;
	std::vector<V3> pos, col;
	char line[1024];
	vertices->clear();
	vcolors->clear();
	vemissions->clear();
	while( std::fgets( line, sizeof( line ), fp ) )
	{
		if( line[0] == 'v' && ( line[1] == ' ' || line[1] == '\t' ) )
		{
			float v[6] = { 0, 0, 0, 1, 1, 1 };
			int n = std::sscanf( line + 2, "%f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5] );
			if( n < 3 ) continue;
			pos.push_back( V3{ v[0], v[1], v[2] } );
			col.push_back( n >= 6 ? V3{ v[3], v[4], v[5] } : V3{ 1, 1, 1 } );
		}
		else if( line[0] == 'f' && ( line[1] == ' ' || line[1] == '\t' ) )
		{
			std::vector<long> idx;
			char* p = line + 2;
			while( *p )
			{
				while( *p == ' ' || *p == '\t' ) p++;
				if( *p == '\0' || *p == '\n' || *p == '\r' ) break;
				char* end = nullptr;
				long i = std::strtol( p, &end, 10 );
				if( end == p ) break;
				if( i < 0 ) i = (long)pos.size() + i; // relative
				else i = i - 1;
				idx.push_back( i );
				p = end;
				while( *p && *p != ' ' && *p != '\t' && *p != '\n' && *p != '\r' ) p++; // skip /t/n
			}
			for( size_t k = 1; k + 1 < idx.size(); k++ )
			{
				const long tri[3] = { idx[0], idx[k], idx[k + 1] };
				bool ok = true;
				for( long t : tri ) ok = ok && t >= 0 && t < (long)pos.size();
				if( !ok ) continue;
				for( long t : tri )
				{
					vertices->push_back( pos[t] );
					vcolors->push_back( col[t] );
					vemissions->push_back( V3{ 0, 0, 0 } );
				}
			}
		}
	}
	std::fclose( fp );
	return !vertices->empty();
}

// getBoundingBox (voxUtil.hpp:66-77) + the dps rule of voxPTGPU.cpp:159-163
inline void boundingGrid( const std::vector<V3>& v, int gridRes, V3* origin, float* dps )
{
	V3 lo = { 3.402823466e+38F, 3.402823466e+38F, 3.402823466e+38F }, hi = { -3.402823466e+38F, -3.402823466e+38F, -3.402823466e+38F };
	for( const V3& p : v )
	{
		lo.x = p.x < lo.x ? p.x : lo.x; lo.y = p.y < lo.y ? p.y : lo.y; lo.z = p.z < lo.z ? p.z : lo.z;
		hi.x = p.x > hi.x ? p.x : hi.x; hi.y = p.y > hi.y ? p.y : hi.y; hi.z = p.z > hi.z ? p.z : hi.z;
	}
	float sx = hi.x - lo.x, sy = hi.y - lo.y, sz = hi.z - lo.z;
	float m = sx > sy ? sx : sy;
	m = m > sz ? m : sz;
	*origin = lo;
	*dps = m / (float)gridRes;
}

inline bool writePpm( const char* path, const uint8_t* rgba, int w, int h )
{
	FILE* fp = std::fopen( path, "wb" );
	if( !fp ) return false;
	std::fprintf( fp, "P6\n%d %d\n255\n", w, h );
	std::vector<uint8_t> row( (size_t)w * 3 );
	for( int y = 0; y < h; y++ )
	{
		for( int x = 0; x < w; x++ )
			for( int c = 0; c < 3; c++ ) row[(size_t)x * 3 + c] = rgba[( (size_t)y * w + x ) * 4 + c];
		std::fwrite( row.data(), 1, row.size(), fp );
	}
	std::fclose( fp );
	return true;
}

// uncompressed PNG (stored deflate blocks), RGBA8
inline bool writePngUncompressed( const char* path, const uint8_t* rgba, int w, int h )
{
	auto crcTable = []() {
		static uint32_t t[256];
		static bool init = false;
		if( !init )
		{
			for( uint32_t n = 0; n < 256; n++ )
			{
				uint32_t c = n;
				for( int k = 0; k < 8; k++ ) c = ( c & 1 ) ? 0xEDB88320u ^ ( c >> 1 ) : c >> 1;
				t[n] = c;
			}
			init = true;
		}
		return t;
	};
	auto crc = [&]( const uint8_t* d, size_t n, uint32_t c ) {
		const uint32_t* t = crcTable();
		for( size_t i = 0; i < n; i++ ) c = t[( c ^ d[i] ) & 0xFF] ^ ( c >> 8 );
		return c;
	};
	FILE* fp = std::fopen( path, "wb" );
	if( !fp ) return false;
	auto be32 = []( uint8_t* p, uint32_t v ) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
	auto chunk = [&]( const char* type, const std::vector<uint8_t>& data ) {
		uint8_t hdr[8];
		be32( hdr, (uint32_t)data.size() );
		std::memcpy( hdr + 4, type, 4 );
		std::fwrite( hdr, 1, 8, fp );
		if( !data.empty() ) std::fwrite( data.data(), 1, data.size(), fp );
		uint32_t c = crc( hdr + 4, 4, 0xFFFFFFFFu );
		c = crc( data.data(), data.size(), c ) ^ 0xFFFFFFFFu;
		uint8_t tail[4];
		be32( tail, c );
		std::fwrite( tail, 1, 4, fp );
	};
	const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	std::fwrite( sig, 1, 8, fp );
	std::vector<uint8_t> ihdr( 13 );
	be32( &ihdr[0], w );
	be32( &ihdr[4], h );
	ihdr[8] = 8;
	ihdr[9] = 6; // RGBA
	chunk( "IHDR", ihdr );
	std::vector<uint8_t> raw;
	raw.reserve( (size_t)h * ( (size_t)w * 4 + 1 ) );
	for( int y = 0; y < h; y++ )
	{
		raw.push_back( 0 );
		raw.insert( raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)( y + 1 ) * w * 4 );
	}
	std::vector<uint8_t> z;
	z.push_back( 0x78 );
	z.push_back( 0x01 );
	uint32_t a = 1, b = 0;
	for( uint8_t v : raw )
	{
		a = ( a + v ) % 65521u;
		b = ( b + a ) % 65521u;
	}
	size_t pos = 0;
	while( pos < raw.size() )
	{
		size_t n = raw.size() - pos;
		if( n > 65535 ) n = 65535;
		z.push_back( pos + n == raw.size() ? 1 : 0 );
		z.push_back( n & 0xFF );
		z.push_back( n >> 8 );
		z.push_back( ~n & 0xFF );
		z.push_back( ( ~n >> 8 ) & 0xFF );
		z.insert( z.end(), raw.begin() + pos, raw.begin() + pos + n );
		pos += n;
	}
	uint8_t ad[4];
	be32( ad, ( b << 16 ) | a );
	z.insert( z.end(), ad, ad + 4 );
	chunk( "IDAT", z );
	chunk( "IEND", std::vector<uint8_t>() );
	std::fclose( fp );
	return true;
}
} // namespace mvrt_io
