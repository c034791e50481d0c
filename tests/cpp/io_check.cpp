// exercises apps/scene_io.hpp for tests/test_apps.py
#include <cstdio>
#include <vector>
#include "scene_io.hpp"
int main( int argc, char** argv )
{
	if( argc < 4 ) return 2;
	std::vector<mvrt_io::V3> v, c, e;
	if( !mvrt_io::readObj( argv[1], &v, &c, &e ) ) return 3;
	double sv = 0, sc = 0;
	for( auto& p : v ) sv += p.x + 2.0 * p.y + 3.0 * p.z;
	for( auto& p : c ) sc += p.x + 2.0 * p.y + 3.0 * p.z;
	mvrt_io::V3 o;
	float dps;
	mvrt_io::boundingGrid( v, 256, &o, &dps );
	std::printf( "%zu %.9g %.9g %.9g %.9g %.9g %.9g\n", v.size(), sv, sc, o.x, o.y, o.z, dps );
	const int W = 37, H = 23;
	std::vector<uint8_t> img( W * H * 4 );
	for( int y = 0; y < H; y++ )
		for( int x = 0; x < W; x++ )
		{
			uint8_t* p = &img[( y * W + x ) * 4];
			p[0] = (uint8_t)( x * 7 );
			p[1] = (uint8_t)( y * 11 );
			p[2] = (uint8_t)( x ^ y );
			p[3] = 255;
		}
	return mvrt_io::writePngUncompressed( argv[2], img.data(), W, H ) && mvrt_io::writePpm( argv[3], img.data(), W, H ) ? 0 : 4;
}
