/*
 * mvrt_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C++ restatement of the reference's algorithm for the hot path named by
 * BASELINE.json (sparse-voxel-octree traversal -> hit shading / bounce -> per-pixel
 * accumulation), plus the host-side pre-steps that produce its inputs (voxelization,
 * voxel merge, DAG build, PMJ table, HDRI summed-area tables).  Every function cites
 * the reference file:line it follows (paths relative to /root/reference).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libmvrt_hip.so) never links, includes or calls it.
 *
 * Pinning status: the reference itself cannot be built in this image without writing
 * stand-in headers (glm, Orochi, intrin.h are absent), so this restatement is pinned by
 *   (1) the reference files that DO compile as they lie (morton.hpp,
 *       libs/smhasher/MurmurHash3.cpp -> oracle/_ref, see oracle/Makefile), and
 *   (2) the golden numbers the survey session captured from the reference source
 *       (tests/golden/survey_appendix_a.json).
 * See DESIGN.md "Oracle".
 *
 * Arithmetic rules: compiled with -ffp-contract=off, no -ffast-math.  mathMode 0 uses
 * libm (the reference's HOST semantics, vectorMath.hpp:93-97); mathMode 1 uses
 * include/mvrt_detmath.h (what the HIP kernels use; bit-exact CPU==GPU).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../include/mvrt_detmath.h"

#define ORC_API extern "C" __attribute__( ( visibility( "default" ) ) )

// ------------------------------------------------------------------------------------
// vector types and helpers (vectorMath.hpp:27-65, 100-248)
// ------------------------------------------------------------------------------------
struct uchar4 { uint8_t x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };

static const float PI = 3.14159265358979323846264338327950288f; // vectorMath.hpp:78
static const float MAX_FLOAT = 3.402823466e+38F;					 // vectorMath.hpp:79

template <class T> static inline T ss_max( T x, T y ) { return ( x < y ) ? y : x; }	   // :100
template <class T> static inline T ss_min( T x, T y ) { return ( y < x ) ? y : x; }	   // :105
template <class T> static inline T ss_abs( T x ) { return x >= T( 0 ) ? x : -x; }	   // :110
template <class T> static inline T ss_clamp( T x, T a, T b ) { return ss_min( ss_max( x, a ), b ); }
static inline float ss_floor( float v ) { return floorf( v ); } // :66 (_mm_floor_ss == floorf)
static inline float ss_ceil( float v ) { return ceilf( v ); }

static inline float2 operator-( float2 a, float2 b ) { return { a.x - b.x, a.y - b.y }; }
static inline float2 operator+( float2 a, float2 b ) { return { a.x + b.x, a.y + b.y }; }
static inline float2 operator*( float2 a, float b ) { return { a.x * b, a.y * b }; }
static inline float dot( float2 a, float2 b ) { return a.x * b.x + a.y * b.y; }
static inline float dot( float3 a, float3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float3 operator-( float3 a ) { return { -a.x, -a.y, -a.z }; }
static inline float3 operator+( float3 a, float3 b ) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline float3 operator-( float3 a, float3 b ) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline float3 operator*( float3 a, float b ) { return { a.x * b, a.y * b, a.z * b }; }
static inline float3 operator*( float a, float3 b ) { return { a * b.x, a * b.y, a * b.z }; }
static inline float3 operator*( float3 a, float3 b ) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
static inline float3 operator/( float3 a, float b ) { return { a.x / b, a.y / b, a.z / b }; }
static inline float3 operator/( float3 a, float3 b ) { return { a.x / b.x, a.y / b.y, a.z / b.z }; }
static inline float3& operator+=( float3& a, float3 b ) { a = a + b; return a; }
static inline float3& operator*=( float3& a, float3 b ) { a = a * b; return a; }
static inline float3 fmaxf3( float3 a, float3 b ) { return { ss_max( a.x, b.x ), ss_max( a.y, b.y ), ss_max( a.z, b.z ) }; }
static inline float3 fminf3( float3 a, float3 b ) { return { ss_min( a.x, b.x ), ss_min( a.y, b.y ), ss_min( a.z, b.z ) }; }
static inline int3 maxi( int3 a, int3 b ) { return { ss_max( a.x, b.x ), ss_max( a.y, b.y ), ss_max( a.z, b.z ) }; }
static inline int3 mini( int3 a, int3 b ) { return { ss_min( a.x, b.x ), ss_min( a.y, b.y ), ss_min( a.z, b.z ) }; }
static inline float3 floorf3( float3 v ) { return { ss_floor( v.x ), ss_floor( v.y ), ss_floor( v.z ) }; }
static inline float3 cross( float3 a, float3 b ) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float3 fabs3( float3 v ) { return { ss_abs( v.x ), ss_abs( v.y ), ss_abs( v.z ) }; }
static inline float maxElement( float a, float b, float c ) { return ss_max( ss_max( a, b ), c ); }
static inline float minElement( float a, float b, float c ) { return ss_min( ss_min( a, b ), c ); }
static inline float mix( float a, float b, float t ) { return a + ( b - a ) * t; }

// math-mode switch (see file header)
struct Math
{
	int mode;
	float cos_( float x ) const { return mode ? mvrt_cos( x ) : cosf( x ); }
	float sin_( float x ) const { return mode ? mvrt_sin( x ) : sinf( x ); }
	float atan2_( float y, float x ) const { return mode ? mvrt_atan2( y, x ) : atan2f( y, x ); }
	float pow_( float x, float y ) const { return mode ? mvrt_pow( x, y ) : powf( x, y ); }
};

// vectorMath.hpp:249-284
static inline float3 closestBarycentricCoordinateOnTriangle( float3 v0, float3 v1, float3 v2, float3 P )
{
	float3 d0 = v0 - P;
	float3 d1 = v1 - P;
	float3 d2 = v2 - P;
	float3 e0 = v2 - v0;
	float3 e1 = v0 - v1;
	float3 e2 = v1 - v2;
	float3 Ng = cross( e2, e0 );
	float U = dot( cross( d2, d0 ), Ng );
	float V = dot( cross( d0, d1 ), Ng );
	float W = dot( cross( d1, d2 ), Ng );
	if( U < 0.0f )
	{
		V = dot( -d0, e0 );
		W = dot( d2, e0 );
	}
	else if( V < 0.0f )
	{
		W = dot( -d1, e1 );
		U = dot( d0, e1 );
	}
	else if( W < 0.0f )
	{
		U = dot( -d2, e2 );
		V = dot( d1, e2 );
	}
	float3 bc = fmaxf3( float3{ 0.0f, 0.0f, 0.0f }, float3{ U, V, W } );
	return bc / ( bc.x + bc.y + bc.z );
}

// ------------------------------------------------------------------------------------
// morton (morton.hpp:5-116; device twin voxKernel.cu:11-44)
// ------------------------------------------------------------------------------------
static inline uint64_t splitBy3( uint32_t a ) // morton.hpp:98-107
{
	uint64_t x = a & 0x1FFFFF;
	x = ( x | x << 32 ) & 0x1f00000000ffffull;
	x = ( x | x << 16 ) & 0x1f0000ff0000ffull;
	x = ( x | x << 8 ) & 0x100f00f00f00f00full;
	x = ( x | x << 4 ) & 0x10c30c30c30c30c3ull;
	x = ( x | x << 2 ) & 0x1249249249249249ull;
	return x;
}
static inline uint64_t mortonEncode( uint32_t x, uint32_t y, uint32_t z ) // morton.hpp:108-113
{
	return splitBy3( x ) | splitBy3( y ) << 1 | splitBy3( z ) << 2;
}
static inline uint32_t getThirdBits( uint64_t m ) // morton.hpp:37-47
{
	uint64_t x = m & 0x1249249249249249ull;
	x = ( x ^ ( x >> 2 ) ) & 0x10c30c30c30c30c3ull;
	x = ( x ^ ( x >> 4 ) ) & 0x100f00f00f00f00full;
	x = ( x ^ ( x >> 8 ) ) & 0x1f0000ff0000ffull;
	x = ( x ^ ( x >> 16 ) ) & 0x1f00000000ffffull;
	x = ( x ^ ( x >> 32 ) ) & 0x1fffffull;
	return (uint32_t)x;
}
ORC_API uint64_t orc_morton_encode( uint32_t x, uint32_t y, uint32_t z ) { return mortonEncode( x, y, z ); }
ORC_API uint64_t orc_morton_encode_naive( uint32_t x, uint32_t y, uint32_t z ) // morton.hpp:5-16
{
	uint64_t code = 0;
	for( uint64_t i = 0; i < 64 / 3; ++i )
	{
		code |= ( (uint64_t)( x & ( 1u << i ) ) << ( 2 * i + 0 ) ) | ( (uint64_t)( y & ( 1u << i ) ) << ( 2 * i + 1 ) ) |
				( (uint64_t)( z & ( 1u << i ) ) << ( 2 * i + 2 ) );
	}
	return code;
}
ORC_API void orc_morton_decode( uint64_t m, uint32_t* xyz ) // morton.hpp:48-53
{
	xyz[0] = getThirdBits( m );
	xyz[1] = getThirdBits( m >> 1 );
	xyz[2] = getThirdBits( m >> 2 );
}
ORC_API void orc_morton_encode_batch( const uint32_t* xyz, int64_t n, uint64_t* out )
{
	for( int64_t i = 0; i < n; i++ ) out[i] = mortonEncode( xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2] );
}

// ------------------------------------------------------------------------------------
// MurmurHash32 / hashCombine (voxCommon.hpp:45-108)
// ------------------------------------------------------------------------------------
static inline uint32_t fmix32( uint32_t h )
{
	h ^= h >> 16;
	h *= 0x85ebca6b;
	h ^= h >> 13;
	h *= 0xc2b2ae35;
	h ^= h >> 16;
	return h;
}
static inline uint32_t rotl32( uint32_t x, int r ) { return ( x << r ) | ( x >> ( 32 - r ) ); }
struct MurmurHash32
{
	uint32_t h1, len;
	MurmurHash32( uint32_t seed ) : h1( seed ), len( 0 ) {}
	void combine( uint32_t k1 )
	{
		k1 *= 0xcc9e2d51;
		k1 = rotl32( k1, 15 );
		k1 *= 0x1b873593;
		h1 ^= k1;
		h1 = rotl32( h1, 13 );
		h1 = h1 * 5 + 0xe6546b64;
		len++;
	}
	uint32_t getHash() const { return fmix32( h1 ^ ( len * 4 ) ); }
};
static inline uint32_t hashCombine( uint32_t a, uint32_t b )
{
	MurmurHash32 h( a );
	h.combine( b );
	return h.getHash();
}
static inline uint32_t hashCombine( uint32_t a, uint32_t b, uint32_t c )
{
	MurmurHash32 h( a );
	h.combine( b );
	h.combine( c );
	return h.getHash();
}
ORC_API uint32_t orc_murmur( uint32_t seed, const uint32_t* words, int n )
{
	MurmurHash32 h( seed );
	for( int i = 0; i < n; i++ ) h.combine( words[i] );
	return h.getHash();
}

// ------------------------------------------------------------------------------------
// PCG32, uniformf (renderCommon.hpp:86-117)
// ------------------------------------------------------------------------------------
struct PCG32
{
	uint64_t state, inc;
	void setup( uint64_t seed, uint64_t stream )
	{
		state = 0;
		inc = stream * 2 + 1;
		nextU32();
		state += seed;
		nextU32();
	}
	uint32_t nextU32()
	{
		uint64_t oldstate = state;
		state = oldstate * 6364136223846793005ULL + inc;
		uint32_t xorshifted = (uint32_t)( ( ( oldstate >> 18u ) ^ oldstate ) >> 27u );
		uint32_t rot = (uint32_t)( oldstate >> 59u );
		return ( xorshifted >> rot ) | ( xorshifted << ( ( -rot ) & 31 ) );
	}
};
static inline float uniformf( uint32_t x )
{
	uint32_t bits = ( x >> 9 ) | 0x3f800000;
	float v;
	memcpy( &v, &bits, 4 );
	return v - 1.0f;
}
ORC_API void orc_pcg32_sequence( uint64_t seed, uint64_t stream, int n, uint32_t* out )
{
	PCG32 r;
	r.setup( seed, stream );
	for( int i = 0; i < n; i++ ) out[i] = r.nextU32();
}
ORC_API float orc_uniformf( uint32_t x ) { return uniformf( x ); }

// ------------------------------------------------------------------------------------
// PMJ02 table + Owen-scrambled lookup (pmjSampler.hpp:14-173)
// ------------------------------------------------------------------------------------
enum { PMJ_LENGTH = 4096, PMJ_N_SEQUENCE = 128 };

// pmjSampler.hpp:14-58.  NOTE: the reference passes uniformFloat(), uniformFloat() as two
// call arguments (:54); C++ leaves their evaluation order unspecified.  The survey's golden
// table hash was produced by clang (left-to-right), which is what we restate: xi0 first.
static void GetPMJ02Samples( int num_samples, float* samples, const std::function<float()>& uniformFloat )
{
	static const uint32_t pmj02_xors[2][32] = {
		{ 0x0, 0x0, 0x2, 0x6, 0x6, 0xe, 0x36, 0x4e, 0x16, 0x2e, 0x276, 0x6ce, 0x716, 0xc2e, 0x3076, 0x40ce, 0x116, 0x22e, 0x20676, 0x60ece, 0x61716,
		  0xe2c2e, 0x367076, 0x4ec0ce, 0x170116, 0x2c022e, 0x2700676, 0x6c00ece, 0x7001716, 0xc002c2e, 0x30007076, 0x4000c0ce },
		{ 0x0, 0x1, 0x3, 0x3, 0x7, 0x1b, 0x27, 0xb, 0x17, 0x13b, 0x367, 0x38b, 0x617, 0x183b, 0x2067, 0x8b, 0x117, 0x1033b, 0x30767, 0x30b8b,
		  0x71617, 0x1b383b, 0x276067, 0xb808b, 0x160117, 0x138033b, 0x3600767, 0x3800b8b, 0x6001617, 0x1800383b, 0x20006067, 0x808b } };
	for( int d = 0; d < 2; d++ ) samples[d] = uniformFloat();
	for( int log_n = 0; ( 1 << log_n ) < num_samples; log_n++ )
	{
		int prev_len = 1 << log_n;
		int n_strata = prev_len * 2;
		float i_strata = 1.0f / n_strata;
		for( int i = 0; i < prev_len && ( prev_len + i ) < num_samples; i++ )
		{
			const int prev_x_idx = i ^ pmj02_xors[0][log_n];
			const int prev_x_stratum = (int)( samples[prev_x_idx * 2] * n_strata );
			const int x_stratum = prev_x_stratum ^ 1;
			const int prev_y_idx = i ^ pmj02_xors[1][log_n];
			const int prev_y_stratum = (int)( samples[prev_y_idx * 2 + 1] * n_strata );
			const int y_stratum = prev_y_stratum ^ 1;
			float* sample = &( samples[( prev_len + i ) * 2] );
			float xi0 = uniformFloat();
			float xi1 = uniformFloat();
			sample[0] = ( xi0 + x_stratum ) * i_strata;
			sample[1] = ( xi1 + y_stratum ) * i_strata;
		}
	}
}
// PMJSampler::setup, pmjSampler.hpp:114-144
ORC_API void orc_pmj_table( float* out /* 2*4096*128 floats */ )
{
	PCG32 rng;
	rng.setup( 0, 2525 );
	for( int i = 0; i < PMJ_N_SEQUENCE; i++ )
	{
		float* p = out + 2 * PMJ_LENGTH * i;
		GetPMJ02Samples( PMJ_LENGTH, p, [&rng]() { return uniformf( rng.nextU32() ); } );
	}
}
static inline uint32_t laine_karras_permutation( uint32_t x, uint32_t seed ) // :62-70
{
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
static inline uint32_t reverseBits( uint32_t v ) // :73-86
{
	v = ( ( v >> 1 ) & 0x55555555 ) | ( ( v & 0x55555555 ) << 1 );
	v = ( ( v >> 2 ) & 0x33333333 ) | ( ( v & 0x33333333 ) << 2 );
	v = ( ( v >> 4 ) & 0x0F0F0F0F ) | ( ( v & 0x0F0F0F0F ) << 4 );
	v = ( ( v >> 8 ) & 0x00FF00FF ) | ( ( v & 0x00FF00FF ) << 8 );
	v = ( v >> 16 ) | ( v << 16 );
	return v;
}
static inline uint32_t nested_uniform_scramble( uint32_t x, uint32_t seed ) // :88-94
{
	x = reverseBits( x );
	x = laine_karras_permutation( x, seed );
	x = reverseBits( x );
	return x;
}
static inline float scramble_f32( float x, uint32_t seed ) // :96-102
{
	x += 1.0f;
	uint32_t bits;
	memcpy( &bits, &x, 4 );
	uint32_t scrambled = 0x3f800000 | ( nested_uniform_scramble( bits & 0x7FFFFF, seed ) & 0x7FFFFF );
	float r;
	memcpy( &r, &scrambled, 4 );
	return r - 1.0f;
}
static inline float2 pmj_sample2d( const float* m_samples, uint32_t sampleIdx, uint32_t dimension, uint32_t stream ) // :158-173
{
	sampleIdx = nested_uniform_scramble( sampleIdx, hashCombine( stream, dimension, 31082745 ) ) & ( PMJ_LENGTH - 1 );
	dimension = nested_uniform_scramble( dimension, hashCombine( stream, 54761983 ) ) & ( PMJ_N_SEQUENCE - 1 );
	uint32_t head = dimension * 2 * PMJ_LENGTH;
	float x = m_samples[head + sampleIdx * 2 + 0];
	float y = m_samples[head + sampleIdx * 2 + 1];
	x = scramble_f32( x, hashCombine( stream, dimension, 83927105 ) );
	y = scramble_f32( y, hashCombine( stream, dimension, 12654890 ) );
	return { x, y };
}
ORC_API uint32_t orc_reverse_bits( uint32_t v ) { return reverseBits( v ); }
ORC_API uint32_t orc_nested_uniform_scramble( uint32_t x, uint32_t seed ) { return nested_uniform_scramble( x, seed ); }
ORC_API float orc_scramble_f32( float x, uint32_t seed ) { return scramble_f32( x, seed ); }
ORC_API void orc_pmj_sample2d( const float* table, uint32_t sampleIdx, uint32_t dim, uint32_t stream, float* out2 )
{
	float2 r = pmj_sample2d( table, sampleIdx, dim, stream );
	out2[0] = r.x;
	out2[1] = r.y;
}

// bSearch, voxCommon.hpp:580-603
ORC_API int orc_bsearch_i32( const int* xs, int n, int x )
{
	int i = 0, j = n;
	while( i < j )
	{
		int m = ( i + j ) / 2;
		int value = xs[m];
		if( value == x ) return m;
		else if( value < x ) i = m + 1;
		else j = m;
	}
	return -1;
}

// ------------------------------------------------------------------------------------
// voxelization (voxelization.hpp:15-337) -- triangle -> voxels, Schwarz & Seidel
// ------------------------------------------------------------------------------------
static inline float2 project2plane( float3 p, int axis ) // :15-34
{
	switch( axis )
	{
	case 0: return { p.x, p.y };
	case 1: return { p.y, p.z };
	default: return { p.z, p.x };
	}
}
static inline float project2plane_reminder( float3 p, int axis ) // :36-48
{
	switch( axis )
	{
	case 0: return p.z;
	case 1: return p.x;
	case 2: return p.y;
	}
	return 0.0f;
}
static inline int2 project2plane( int3 p, int axis ) // :50-69
{
	switch( axis )
	{
	case 0: return { p.x, p.y };
	case 1: return { p.y, p.z };
	default: return { p.z, p.x };
	}
}
static inline int project2plane_reminder( int3 p, int axis ) // :70-82
{
	switch( axis )
	{
	case 0: return p.z;
	case 1: return p.x;
	case 2: return p.y;
	}
	return 0;
}
static inline int majorAxis( float3 d ) // :84-94
{
	float x = ss_abs( d.x ), y = ss_abs( d.y ), z = ss_abs( d.z );
	if( x < y ) return y < z ? 0 : 2;
	return x < z ? 0 : 1;
}
static inline float3 unProjectPlane( float2 p, float reminder, int axis ) // :96-108
{
	switch( axis )
	{
	case 0: return { p.x, p.y, reminder };
	case 1: return { reminder, p.x, p.y };
	case 2: return { p.y, reminder, p.x };
	}
	return { 0.0f, 0.0f, 0.0f };
}
static inline int3 unProjectPlane( int2 p, int reminder, int axis ) // :110-122
{
	switch( axis )
	{
	case 0: return { p.x, p.y, reminder };
	case 1: return { reminder, p.x, p.y };
	case 2: return { p.y, reminder, p.x };
	}
	return { 0, 0, 0 };
}

struct VTContext // :124-337
{
	int major;
	int2 lower_xy, upper_xy;
	int lower_z, upper_z;
	float d_consts[3][3];
	float nesx[3][3];
	float nesy[3][3];
	float2 origin_xy;
	float origin_z;
	float kx, ky, constant_max, constant_min, constant_six;

	VTContext( float3 v0, float3 v1, float3 v2, bool sixSeparating, float3 origin, float dps, int gridRes ) // :148-229
	{
		float3 e01 = v1 - v0;
		float3 e12 = v2 - v1;
		float3 n = cross( e01, e12 );
		major = majorAxis( n );

		float3 bbox_lower = fminf3( fminf3( v0, v1 ), v2 );
		float3 bbox_upper = fmaxf3( fmaxf3( v0, v1 ), v2 );
		float3 lowerf = floorf3( ( bbox_lower - origin ) / dps );
		float3 upperf = floorf3( ( bbox_upper - origin ) / dps );
		int3 lower = { (int)lowerf.x, (int)lowerf.y, (int)lowerf.z };
		int3 upper = { (int)upperf.x, (int)upperf.y, (int)upperf.z };
		lower = maxi( lower, int3{ 0, 0, 0 } );
		upper = mini( upper, int3{ gridRes - 1, gridRes - 1, gridRes - 1 } );

		lower_xy = project2plane( lower, major );
		upper_xy = project2plane( upper, major );
		lower_z = project2plane_reminder( lower, major );
		upper_z = project2plane_reminder( upper, major );

		for( int axis = 0; axis < 3; axis++ )
		{
			float2 dp_proj = float2{ dps, dps };
			float2 vs_proj[3] = { project2plane( v0, axis ), project2plane( v1, axis ), project2plane( v2, axis ) };
			float reminder = project2plane_reminder( n, axis );
			float n_sign = 0.0f < reminder ? 1.0f : -1.0f;
			for( int edge = 0; edge < 3; edge++ )
			{
				float2 a = vs_proj[edge];
				float2 b = vs_proj[( edge + 1 ) % 3];
				float2 e = b - a;
				float2 ne = float2{ -e.y, e.x } * n_sign;
				nesx[axis][edge] = ne.x;
				nesy[axis][edge] = ne.y;
				float d_const;
				if( sixSeparating == false )
				{
					d_const = ss_max( ne.x * dp_proj.x, 0.0f ) + ss_max( ne.y * dp_proj.y, 0.0f ) - dot( ne, a );
				}
				else
				{
					d_const = dot( ne, dp_proj * 0.5f - a ) + 0.5f * dps * ss_max( ss_abs( ne.x ), ss_abs( ne.y ) );
				}
				d_consts[axis][edge] = d_const;
			}
		}
		origin_xy = project2plane( origin, major );
		origin_z = project2plane_reminder( origin, major );
		float2 v0_xy = project2plane( v0, major );
		float v0_z = project2plane_reminder( v0, major );
		float2 n_xy = project2plane( n, major );
		float n_z = project2plane_reminder( n, major );
		kx = -n_xy.x / n_z;
		ky = -n_xy.y / n_z;
		float K = -kx * v0_xy.x - ky * v0_xy.y + v0_z;
		constant_max = K + dps * ( ss_max( kx, 0.0f ) + ss_max( ky, 0.0f ) );
		constant_min = K + dps * ( ss_min( kx, 0.0f ) + ss_min( ky, 0.0f ) );
		constant_six = K + 0.5f * dps * ( kx + ky );
	}
	int2 xRangeInclusive() const { return { lower_xy.x, upper_xy.x }; } // :230-233
	int2 yRangeInclusive( int x, float dps ) const						// :235-274
	{
		float xcoord = origin_xy.x + x * dps;
		float miny = -3.402823466e+38F;
		float maxy = 3.402823466e+38F;
		for( int edge = 0; edge < 3; edge++ )
		{
			float nex = nesx[major][edge];
			float ney = nesy[major][edge];
			float d_const = d_consts[major][edge];
			if( ney == 0.0f )
			{
				if( -nex * xcoord <= d_const ) continue;
				else return { 1, -1 };
			}
			float k = -( xcoord * nex + d_const ) / ney;
			if( 0.0f < ney ) miny = ss_max( miny, k );
			else maxy = ss_min( maxy, k );
		}
		float minIndexF = ss_max( ( miny - origin_xy.y ) / dps, -2147483648.0f );
		float maxIndexF = ss_min( ( maxy - origin_xy.y ) / dps, 2147483520.0f );
		int lowerY = (int)ss_ceil( minIndexF );
		int upperY = (int)ss_floor( maxIndexF );
		lowerY = ss_max( lowerY, lower_xy.y );
		upperY = ss_min( upperY, upper_xy.y );
		return { lowerY, upperY };
	}
	int2 zRangeInclusive( int x, int y, float dps, bool sixSeparating ) const // :275-306
	{
		float2 o_xy = origin_xy + float2{ dps * x, dps * y };
		float var = kx * o_xy.x + ky * o_xy.y;
		int zmin, zmax;
		if( sixSeparating )
		{
			float tsix = var + constant_six;
			float indexf = ( tsix - origin_z ) / dps;
			float zf = ss_floor( indexf );
			int z = (int)zf;
			zmin = indexf == zf ? z - 1 : z;
			zmax = z;
		}
		else
		{
			float tmax = var + constant_max;
			float tmin = var + constant_min;
			zmin = (int)( ss_floor( ( tmin - origin_z ) / dps ) );
			zmax = (int)( ss_floor( ( tmax - origin_z ) / dps ) );
		}
		zmin = ss_max( zmin, lower_z );
		zmax = ss_min( zmax, upper_z );
		return { zmin, zmax };
	}
	float3 p( int x, int y, int z, float dps ) const // :307-312
	{
		float2 p_proj = origin_xy + float2{ dps * x, dps * y };
		float reminder = origin_z + (float)z * dps;
		return unProjectPlane( p_proj, reminder, major );
	}
	int3 i( int x, int y, int z ) const { return unProjectPlane( int2{ x, y }, z, major ); } // :313-316
	bool intersect( float3 p ) const														 // :318-336
	{
		if( lower_z == upper_z ) return true;
		for( int axis = 0; axis < 3; axis++ )
		{
			if( axis == major ) continue;
			float2 p_proj = project2plane( p, axis );
			for( int edge = 0; edge < 3; edge++ )
			{
				float nex = nesx[axis][edge];
				float ney = nesy[axis][edge];
				float d = nex * p_proj.x + ney * p_proj.y + d_consts[axis][edge];
				if( d < 0.0f ) return false;
			}
		}
		return true;
	}
};

struct VoxelAttirb // voxCommon.hpp:121-125
{
	uchar4 color;
	uchar4 emission;
};

// voxRT.cpp:198-240 (CPU voxelize loop) == voxKernel.cu:91-166 (voxelize kernel body).
// tris/colors/emissions: nTri*9 floats.  Returns the number of dumped voxels (duplicates kept);
// fills outputs up to `capacity` (pass 0/null to count only).
ORC_API int64_t orc_voxelize( const float* tris, const float* cols, const float* emis, int64_t nTri, const float* origin3, float dps, int gridRes,
							  int sixSeparating, uint64_t* mortonOut, uint8_t* attrOut, int64_t capacity )
{
	int64_t n = 0;
	float3 origin = { origin3[0], origin3[1], origin3[2] };
	for( int64_t t = 0; t < nTri; t++ )
	{
		const float* v = tris + t * 9;
		float3 v0 = { v[0], v[1], v[2] }, v1 = { v[3], v[4], v[5] }, v2 = { v[6], v[7], v[8] };
		float3 c0 = { 1, 1, 1 }, c1 = c0, c2 = c0, e0 = { 0, 0, 0 }, e1 = e0, e2 = e0;
		if( cols )
		{
			const float* c = cols + t * 9;
			c0 = { c[0], c[1], c[2] };
			c1 = { c[3], c[4], c[5] };
			c2 = { c[6], c[7], c[8] };
		}
		if( emis )
		{
			const float* e = emis + t * 9;
			e0 = { e[0], e[1], e[2] };
			e1 = { e[3], e[4], e[5] };
			e2 = { e[6], e[7], e[8] };
		}
		VTContext context( v0, v1, v2, sixSeparating != 0, origin, dps, gridRes );
		int2 xrange = context.xRangeInclusive();
		for( int x = xrange.x; x <= xrange.y; x++ )
		{
			int2 yrange = context.yRangeInclusive( x, dps );
			for( int y = yrange.x; y <= yrange.y; y++ )
			{
				int2 zrange = context.zRangeInclusive( x, y, dps, sixSeparating != 0 );
				for( int z = zrange.x; z <= zrange.y; z++ )
				{
					float3 p = context.p( x, y, z, dps );
					if( context.intersect( p ) )
					{
						if( n < capacity && mortonOut )
						{
							int3 c = context.i( x, y, z );
							mortonOut[n] = mortonEncode( c.x, c.y, c.z );
							float3 bc = closestBarycentricCoordinateOnTriangle( v0, v1, v2, p );
							float3 bColor = bc.x * c1 + bc.y * c2 + bc.z * c0;
							float3 bEmission = bc.x * e1 + bc.y * e2 + bc.z * e0;
							uint8_t* a = attrOut + n * 8;
							a[0] = (uint8_t)( bColor.x * 255.0f + 0.5f );
							a[1] = (uint8_t)( bColor.y * 255.0f + 0.5f );
							a[2] = (uint8_t)( bColor.z * 255.0f + 0.5f );
							a[3] = 255;
							a[4] = (uint8_t)( bEmission.x * 255.0f + 0.5f );
							a[5] = (uint8_t)( bEmission.y * 255.0f + 0.5f );
							a[6] = (uint8_t)( bEmission.z * 255.0f + 0.5f );
							a[7] = 255;
						}
						n++;
					}
				}
			}
		}
	}
	return n;
}

// mergeVoxels (voxRT.cpp:14-51) == `unique` kernel (voxKernel.cu:170-243): sort by morton, one
// entry per distinct code, colour/emission = INTEGER mean over duplicates, alpha 255.
// In place; returns the unique count and *hasEmission (voxKernel.cu:225-228).
ORC_API int64_t orc_merge_voxels( uint64_t* morton, uint8_t* attrs, int64_t n, int* hasEmission )
{
	std::vector<int64_t> order( n );
	for( int64_t i = 0; i < n; i++ ) order[i] = i;
	std::stable_sort( order.begin(), order.end(), [&]( int64_t a, int64_t b ) { return morton[a] < morton[b]; } );
	std::vector<uint64_t> m2;
	std::vector<uint8_t> a2;
	int he = 0;
	for( int64_t i = 0; i < n; )
	{
		uint64_t key = morton[order[i]];
		int sum[6] = { 0, 0, 0, 0, 0, 0 };
		int cnt = 0;
		int64_t j = i;
		for( ; j < n && morton[order[j]] == key; j++ )
		{
			const uint8_t* a = attrs + order[j] * 8;
			sum[0] += a[0];
			sum[1] += a[1];
			sum[2] += a[2];
			sum[3] += a[4];
			sum[4] += a[5];
			sum[5] += a[6];
			cnt++;
		}
		uint8_t o[8] = { (uint8_t)( sum[0] / cnt ), (uint8_t)( sum[1] / cnt ), (uint8_t)( sum[2] / cnt ), 255,
						 (uint8_t)( sum[3] / cnt ), (uint8_t)( sum[4] / cnt ), (uint8_t)( sum[5] / cnt ), 255 };
		if( o[4] || o[5] || o[6] ) he = 1;
		m2.push_back( key );
		a2.insert( a2.end(), o, o + 8 );
		i = j;
	}
	memcpy( morton, m2.data(), m2.size() * 8 );
	memcpy( attrs, a2.data(), a2.size() );
	if( hasEmission ) *hasEmission = he;
	return (int64_t)m2.size();
}

// ------------------------------------------------------------------------------------
// octree / DAG (voxCommon.hpp:29-36,133-195; IntersectorOctree.hpp:11-123,241-247)
// ------------------------------------------------------------------------------------
struct OctreeNode // voxCommon.hpp:133-138 ; sizeof == 68
{
	uint8_t mask;
	uint32_t children[8];
	uint32_t nVoxelsPSum[8];
	bool operator<( const OctreeNode& rhs ) const // :163-180
	{
		if( mask != rhs.mask ) return mask < rhs.mask;
		for( int i = 0; i < 8; i++ )
		{
			if( children[i] == rhs.children[i] ) continue;
			return children[i] < rhs.children[i];
		}
		return false;
	}
};
static_assert( sizeof( OctreeNode ) == 68, "reference OctreeNode is 68 bytes" );
struct OctreeTask // voxCommon.hpp:29-36
{
	uint64_t morton;
	uint32_t child;
	uint32_t numberOfVoxels;
	uint64_t getMortonParent() const { return morton >> 3; }
};

// buildOctreeDAGReference, IntersectorOctree.hpp:11-123 (dag != 0), or the same without the
// std::map dedup (dag == 0: what ENABLE_GPU_DAG-off produces, voxKernel.cu:322-334, with valid
// nVoxelsPSum).  Then embedMasks (IntersectorOctree.hpp:241-247, voxCommon.hpp:183-195) if
// embed != 0.  Returns node count; nodesOut may be null to count.  Root = last node.
ORC_API int64_t orc_build_octree( const uint64_t* mortonVoxels, int64_t nVoxels, int wide, int dag, int embed, uint8_t* nodesOut, int64_t capacityNodes )
{
	std::vector<OctreeNode> nodes;
	std::vector<OctreeTask> curTasks( nVoxels ), nextTasks;
	for( int64_t i = 0; i < nVoxels; i++ )
	{
		curTasks[i].morton = mortonVoxels[i];
		curTasks[i].child = 0xFFFFFFFFu;
		curTasks[i].numberOfVoxels = 1;
	}
	std::map<OctreeNode, int> existings;
	while( 1 < wide )
	{
		size_t i = 0;
		while( i < curTasks.size() )
		{
			size_t beg = i;
			uint64_t pMorton = curTasks[beg].getMortonParent();
			size_t end = beg + 1;
			while( end < curTasks.size() && curTasks[end].getMortonParent() == pMorton ) end++;

			OctreeNode node;
			memset( &node, 0, sizeof( node ) );
			node.mask = 0;
			for( int j = 0; j < 8; j++ )
			{
				node.children[j] = 0xFFFFFFFFu;
				node.nVoxelsPSum[j] = 0;
			}
			for( size_t j = beg; j < end; j++ )
			{
				uint32_t space = curTasks[j].morton & 0x7;
				node.mask |= ( 1 << space ) & 0xFF;
				node.children[space] = curTasks[j].child;
				node.nVoxelsPSum[space] = curTasks[j].numberOfVoxels;
			}
			int numberOfVoxels = 0;
			for( int j = 0; j < 8; j++ )
			{
				uint32_t c = node.nVoxelsPSum[j];
				node.nVoxelsPSum[j] = numberOfVoxels;
				numberOfVoxels += c;
			}
			uint32_t nodeIndex;
			if( dag )
			{
				auto it = existings.find( node );
				if( it == existings.end() )
				{
					nodeIndex = (uint32_t)nodes.size();
					nodes.push_back( node );
					existings[node] = nodeIndex;
				}
				else
				{
					nodeIndex = it->second;
				}
			}
			else
			{
				nodeIndex = (uint32_t)nodes.size();
				nodes.push_back( node );
			}
			OctreeTask nextTask;
			nextTask.morton = pMorton;
			nextTask.child = nodeIndex;
			nextTask.numberOfVoxels = numberOfVoxels;
			nextTasks.push_back( nextTask );
			i = end;
		}
		curTasks.clear();
		std::swap( curTasks, nextTasks );
		wide /= 2;
	}
	if( embed )
	{
		for( size_t i = 0; i < nodes.size(); i++ )
		{
			for( int k = 0; k < 8; k++ )
			{
				uint32_t child = nodes[i].children[k];
				if( child == 0xFFFFFFFFu ) continue;
				child |= (uint32_t)nodes[child].mask << 24;
				nodes[i].children[k] = child;
			}
		}
	}
	if( nodesOut )
	{
		int64_t n = std::min<int64_t>( capacityNodes, (int64_t)nodes.size() );
		memcpy( nodesOut, nodes.data(), n * sizeof( OctreeNode ) );
	}
	return (int64_t)nodes.size();
}

// ------------------------------------------------------------------------------------
// traversal (voxCommon.hpp:201-423, SMALL_STACK + ENABLE_EMBEDED_MASK variant; the
// non-embedded variant :353-356 selected by embedded == 0)
// ------------------------------------------------------------------------------------
struct StackElement // voxCommon.hpp:202-212
{
	uint32_t nodeIndex;
	float tx1, ty1, tz1;
	float scale;
	uint32_t childMask;
	uint32_t nVoxelSkipped;
	uint32_t _pad;
};

struct TraceStats
{
	uint64_t descents;
	int maxSp;
};
// traversal event tallies over every ray traced with stats (diagnostics for the kernel design, read by orc_trace_events):
// [0] rays, [1] candidate tests (inner-loop turns), [2] descents, [3] pushes, [4] pops, [5] pops that were followed by a descent
// from the restored node before the next pop ("fruitful"), [6] leaf checks, [7] node visits (outer-loop turns on inner nodes)
static std::atomic<uint64_t> g_events[8];
static std::atomic<int> g_eventsOn{ 0 }; // off by default: 8 contended atomics per ray cost the 16-thread CPU baseline a factor of 3

static void octreeTraverse_EfficientParametric( const OctreeNode* nodes, uint32_t nodeIndex, StackElement* stack, float3 ro, float3 rd, const float3& lower,
												const float3& upper, float* t, int* nMajor, uint32_t* vIndex, bool isShadowRay, bool embedded, TraceStats* stats )
{
	float3 one_over_rd = float3{ 1.0f, 1.0f, 1.0f } / rd; // :240
	uint32_t vMask = 0;
	if( one_over_rd.x < 0.0f ) // :243-260
	{
		vMask |= 1u;
		one_over_rd.x = -one_over_rd.x;
		ro.x = lower.x + upper.x - ro.x;
	}
	if( one_over_rd.y < 0.0f )
	{
		vMask |= 2u;
		one_over_rd.y = -one_over_rd.y;
		ro.y = lower.y + upper.y - ro.y;
	}
	if( one_over_rd.z < 0.0f )
	{
		vMask |= 4u;
		one_over_rd.z = -one_over_rd.z;
		ro.z = lower.z + upper.z - ro.z;
	}
	float3 bound = float3{ MAX_FLOAT, MAX_FLOAT, MAX_FLOAT } / fmaxf3( fmaxf3( fabs3( lower - ro ), fabs3( upper - ro ) ), float3{ 1.0f, 1.0f, 1.0f } ); // :265-268
	one_over_rd = fminf3( one_over_rd, bound );																										  // :269
	float3 t0 = ( lower - ro ) * one_over_rd;																										  // :271
	float3 t1 = ( upper - ro ) * one_over_rd;
	float S_lmaxTop = maxElement( t0.x, t0.y, t0.z );
	if( minElement( t1.x, t1.y, t1.z ) < S_lmaxTop ) return; // :275-278

	if( embedded ) nodeIndex |= (uint32_t)nodes[nodeIndex].mask << 24; // :306

	int sp = 0;
	uint64_t ev[8] = { 1, 0, 0, 0, 0, 0, 0, 0 };
	bool justPopped = false;
	StackElement cur = { nodeIndex, t1.x, t1.y, t1.z, 1.0f, 0xFFFFFFFF, 0, 0 }; // :310
	float3 dt = t1 - t0;													   // :312

	for( ;; )
	{
		// next: (:316)
		float tx0 = cur.tx1 - dt.x * cur.scale;
		float ty0 = cur.ty1 - dt.y * cur.scale;
		float tz0 = cur.tz1 - dt.z * cur.scale;
		float S_lmax = maxElement( tx0, ty0, tz0 );

		bool descended = false;
		if( cur.nodeIndex == 0xFFFFFFFFu ) // :322
		{
			ev[6]++;
			if( 0.0f < S_lmax )
			{
				*t = S_lmax;
				*nMajor = S_lmax == tx0 ? 1 : ( S_lmax == ty0 ? 2 : 0 );
				*vIndex = cur.nVoxelSkipped;
				break;
			}
			// goto pop
		}
		else
		{
			ev[7]++;
			float txM = 0.5f * ( tx0 + cur.tx1 ); // :338-340
			float tyM = 0.5f * ( ty0 + cur.ty1 );
			float tzM = 0.5f * ( tz0 + cur.tz1 );
			if( cur.childMask == 0xFFFFFFFF ) // :342-348
			{
				cur.childMask = ( txM < S_lmax ? 1u : 0u ) | ( tyM < S_lmax ? 2u : 0u ) | ( tzM < S_lmax ? 4u : 0u );
			}
			uint32_t mask;
			const OctreeNode* node;
			if( embedded ) // :350-356
			{
				mask = cur.nodeIndex >> 24;
				node = &nodes[cur.nodeIndex & 0xFFFFFF];
			}
			else
			{
				node = &nodes[cur.nodeIndex];
				mask = node->mask;
			}
			float x1 = ( cur.childMask & 1u ) ? cur.tx1 : txM; // :358-360
			float y1 = ( cur.childMask & 2u ) ? cur.ty1 : tyM;
			float z1 = ( cur.childMask & 4u ) ? cur.tz1 : tzM;
			for( ;; ) // :362-412
			{
				ev[1]++;
				float S_umin_next = minElement( x1, y1, z1 );
				uint32_t mv = S_umin_next == x1 ? 1u : ( S_umin_next == y1 ? 2u : 4u );
				bool hasNext = ( cur.childMask & mv ) == 0;
				uint32_t childIndex = cur.childMask ^ vMask;
				cur.childMask |= mv;
				bool isBehind = S_umin_next < 0.0f;
				if( ( mask & ( 0x1 << childIndex ) ) && isBehind == false )
				{
					if( hasNext )
					{
						stack[sp++] = cur;
						ev[3]++;
						if( stats && sp > stats->maxSp ) stats->maxSp = sp;
					}
					cur.nodeIndex = node->children[childIndex];
					if( stats ) stats->descents++;
					ev[2]++;
					if( justPopped ) ev[5]++;
					justPopped = false;
					cur.tx1 = x1;
					cur.ty1 = y1;
					cur.tz1 = z1;
					cur.scale *= 0.5f;
					cur.childMask = 0xFFFFFFFF;
					if( isShadowRay == false )
					{
						cur.nVoxelSkipped += node->nVoxelsPSum[childIndex];
					}
					descended = true;
					break; // goto next
				}
				if( hasNext == false ) break;
				switch( mv )
				{
				case 1: x1 = cur.tx1; break;
				case 2: y1 = cur.ty1; break;
				case 4: z1 = cur.tz1; break;
				}
			}
		}
		if( descended ) continue;
		// pop: (:414-422)
		if( sp )
		{
			cur = stack[--sp];
			ev[4]++;
			justPopped = true;
		}
		else break;
	}
	if( stats && g_eventsOn.load( std::memory_order_relaxed ) )
		for( int k = 0; k < 8; k++ ) g_events[k].fetch_add( ev[k], std::memory_order_relaxed );
}
ORC_API void orc_trace_events_enable( int on ) { g_eventsOn.store( on ); }
ORC_API void orc_trace_events( uint64_t out[8], int reset )
{
	for( int k = 0; k < 8; k++ )
	{
		out[k] = g_events[k].load();
		if( reset ) g_events[k].store( 0 );
	}
}

struct Scene
{
	std::vector<OctreeNode> nodes;
	std::vector<VoxelAttirb> attrs;
	float3 lower, upper;
	float dps;
	float emissionScale = 7.5f; // IntersectorOctreeGPU.hpp:273
	uint32_t hasEmission = 0;
	bool embedded = true;

	// IntersectorOctreeGPU::intersect, IntersectorOctreeGPU.hpp:243-251 (root = last node)
	void intersect( StackElement* stack, float3 ro, float3 rd, float* t, int* nMajor, uint32_t* vIndex, bool isShadowRay, TraceStats* st ) const
	{
		octreeTraverse_EfficientParametric( nodes.data(), (uint32_t)nodes.size() - 1, stack, ro, rd, lower, upper, t, nMajor, vIndex, isShadowRay, embedded, st );
	}
	uchar4 getVoxelColor( uint32_t vIndex ) const { return attrs[vIndex].color; } // :252-255
	float3 getVoxelEmission( uint32_t vIndex, bool withScale ) const				  // :256-259
	{
		uchar4 e = attrs[vIndex].emission;
		float3 r = { (float)e.x / 255.0f, (float)e.y / 255.0f, (float)e.z / 255.0f };
		return r * ( withScale ? emissionScale : 1.0f );
	}
};

ORC_API void* orc_scene_create( const uint8_t* nodes68, int64_t nNodes, const uint8_t* attrs8, int64_t nVoxels, const float* origin3, float dps, int gridRes, int hasEmission,
								int embedded )
{
	Scene* s = new Scene();
	s->nodes.resize( nNodes );
	memcpy( s->nodes.data(), nodes68, nNodes * sizeof( OctreeNode ) );
	s->attrs.resize( std::max<int64_t>( nVoxels, 1 ) );
	if( nVoxels ) memcpy( s->attrs.data(), attrs8, nVoxels * 8 );
	// IntersectorOctreeGPU.hpp:78-80
	s->lower = { origin3[0], origin3[1], origin3[2] };
	s->upper = s->lower + float3{ dps, dps, dps } * (float)gridRes;
	s->dps = dps;
	s->hasEmission = hasEmission;
	s->embedded = embedded != 0;
	return s;
}
ORC_API void orc_scene_destroy( void* s ) { delete (Scene*)s; }
ORC_API void orc_scene_set_emission_scale( void* s, float scale ) { ( (Scene*)s )->emissionScale = scale; } // m_emissionScale, IntersectorOctreeGPU.hpp:273
ORC_API void orc_scene_bounds( void* sp, float* lower3, float* upper3 )
{
	Scene* s = (Scene*)sp;
	lower3[0] = s->lower.x; lower3[1] = s->lower.y; lower3[2] = s->lower.z;
	upper3[0] = s->upper.x; upper3[1] = s->upper.y; upper3[2] = s->upper.z;
}

static void parallelFor( int64_t n, int nThreads, const std::function<void( int64_t, int64_t, int )>& body )
{
	if( nThreads <= 1 )
	{
		body( 0, n, 0 );
		return;
	}
	std::vector<std::thread> th;
	std::atomic<int64_t> next( 0 );
	const int64_t chunk = std::max<int64_t>( 1, std::min<int64_t>( 4096, n / ( nThreads * 8 ) + 1 ) );
	for( int k = 0; k < nThreads; k++ )
	{
		th.emplace_back( [&, k]() {
			for( ;; )
			{
				int64_t b = next.fetch_add( chunk );
				if( b >= n ) break;
				body( b, std::min( n, b + chunk ), k );
			}
		} );
	}
	for( auto& t : th ) t.join();
}

// Batched form of IntersectorOctree::intersect (IntersectorOctree.hpp:248-257).
// ro/rd: n*3 floats (AoS).  isShadow: per-ray flags or null (all false).  Outputs keep the
// reference conventions: t stays MAX_FLOAT on a miss, nMajor/vIndex are then left at the
// initial values the callers use (nMajor = -1 here for determinism, vIndex = 0).
// descents (optional) = child-pointer fetches per ray (voxCommon.hpp:381).
ORC_API void orc_trace_batch( void* sp, int64_t n, const float* ro, const float* rd, const uint8_t* isShadow, float* tOut, int32_t* nMajorOut, uint32_t* vIndexOut,
							  uint32_t* descentsOut, int32_t* maxSpOut, int nThreads )
{
	Scene* s = (Scene*)sp;
	std::vector<int> maxSps( std::max( nThreads, 1 ), 0 );
	parallelFor( n, nThreads, [&]( int64_t b, int64_t e, int k ) {
		StackElement stack[32]; // IntersectorOctree.hpp:250
		for( int64_t i = b; i < e; i++ )
		{
			float t = MAX_FLOAT;
			int nMajor = -1;
			uint32_t vIndex = 0;
			TraceStats st = { 0, 0 };
			s->intersect( stack, float3{ ro[i * 3], ro[i * 3 + 1], ro[i * 3 + 2] }, float3{ rd[i * 3], rd[i * 3 + 1], rd[i * 3 + 2] }, &t, &nMajor, &vIndex,
						  isShadow ? isShadow[i] != 0 : false, &st );
			tOut[i] = t;
			if( nMajorOut ) nMajorOut[i] = nMajor;
			if( vIndexOut ) vIndexOut[i] = vIndex;
			if( descentsOut ) descentsOut[i] = (uint32_t)st.descents;
			if( st.maxSp > maxSps[k] ) maxSps[k] = st.maxSp;
		}
	} );
	if( maxSpOut )
	{
		int m = 0;
		for( int v : maxSps ) m = std::max( m, v );
		*maxSpOut = m;
	}
}

// getHitN, voxCommon.hpp:564-577
static inline float3 getHitN( int major, float3 rd )
{
	switch( major )
	{
	case 0: return { 0.0f, 0.0f, 0.0f < rd.z ? -1.0f : 1.0f };
	case 1: return { 0.0f < rd.x ? -1.0f : 1.0f, 0.0f, 0.0f };
	case 2: return { 0.0f, 0.0f < rd.y ? -1.0f : 1.0f, 0.0f };
	}
	return { 0.0f, 0.0f, 0.0f };
}

// ------------------------------------------------------------------------------------
// camera (renderCommon.hpp:17-84)
// ------------------------------------------------------------------------------------
struct CameraPinhole
{
	float3 m_o, m_front, m_up, m_right;
	float m_tanHthetaY, m_lensR, m_focus;
	// :21-35.  view/proj are column-major 4x4 (glm): M[c][r] = m[c*4+r]
	void initFromPerspective( const float* view, const float* proj, float focus, float lensR )
	{
		// vT = transpose(mat3(view)); vT[i] (column i of vT) = row i of mat3(view)
		float3 r0 = { view[0], view[4], view[8] };
		float3 r1 = { view[1], view[5], view[9] };
		float3 r2 = { view[2], view[6], view[10] };
		m_front = { -r2.x, -r2.y, -r2.z };
		m_up = r1;
		m_right = r0;
		// m = vT * vec3(view[3]) = r0 * v.x + r1 * v.y + r2 * v.z  (glm mat*vec column combination order)
		float3 v = { view[12], view[13], view[14] };
		float3 m = { r0.x * v.x + r1.x * v.y + r2.x * v.z, r0.y * v.x + r1.y * v.y + r2.y * v.z, r0.z * v.x + r1.z * v.y + r2.z * v.z };
		m_o = { -m.x, -m.y, -m.z };
		m_tanHthetaY = 1.0f / proj[5];
		m_lensR = lensR;
		m_focus = focus;
	}
	void shoot( float3* ro, float3* rd, int x, int y, float xo, float yo, int imageWidth, int imageHeight ) const // :37-49
	{
		float xf = ( x + xo ) / imageWidth;
		float yf = ( y + yo ) / imageHeight;
		float3 d = m_right * mix( -m_tanHthetaY, m_tanHthetaY, xf ) * imageWidth / imageHeight + m_up * mix( m_tanHthetaY, -m_tanHthetaY, yf ) + m_front;
		*ro = m_o;
		*rd = d;
	}
	void shootThinLens( float3* ro, float3* rd, int x, int y, float xo, float yo, int imageWidth, int imageHeight, float u0, float u1 ) const // :50-75
	{
		float xf = ( x + xo ) / imageWidth;
		float yf = ( y + yo ) / imageHeight;
		float3 focalP = { m_focus * mix( -m_tanHthetaY, m_tanHthetaY, xf ) * imageWidth / imageHeight, m_focus * mix( m_tanHthetaY, -m_tanHthetaY, yf ), m_focus };
		float3 lensP = { mix( -m_lensR, m_lensR, u0 ), mix( -m_lensR, m_lensR, u1 ), 0.0f };
		float3 dir = focalP - lensP;
		float3 d = m_right * dir.x + m_up * dir.y + m_front * dir.z;
		*rd = d;
		*ro = m_o + m_right * lensP.x + m_up * lensP.y + m_front * lensP.z;
	}
};
static_assert( sizeof( CameraPinhole ) == 60, "reference CameraPinhole is 60 bytes" );

ORC_API void orc_camera_from_matrices( const float* view16, const float* proj16, float focus, float lensR, float* cam15 )
{
	CameraPinhole c;
	c.initFromPerspective( view16, proj16, focus, lensR );
	memcpy( cam15, &c, sizeof( c ) );
}
ORC_API void orc_camera_shoot( const float* cam15, int x, int y, float xo, float yo, int W, int H, int thinLens, float u0, float u1, float* ro3, float* rd3 )
{
	CameraPinhole c;
	memcpy( &c, cam15, sizeof( c ) );
	float3 ro, rd;
	if( thinLens ) c.shootThinLens( &ro, &rd, x, y, xo, yo, W, H, u0, u1 );
	else c.shoot( &ro, &rd, x, y, xo, yo, W, H );
	ro3[0] = ro.x; ro3[1] = ro.y; ro3[2] = ro.z;
	rd3[0] = rd.x; rd3[1] = rd.y; rd3[2] = rd.z;
}

// ------------------------------------------------------------------------------------
// sampling helpers (renderCommon.hpp:119-202)
// ------------------------------------------------------------------------------------
static inline void GetOrthonormalBasis( float3 zaxis, float3* xaxis, float3* yaxis ) // :119-126
{
	const float sign = copysignf( 1.0f, zaxis.z );
	const float a = -1.0f / ( sign + zaxis.z );
	const float b = zaxis.x * zaxis.y * a;
	*xaxis = float3{ 1.0f + sign * zaxis.x * zaxis.x * a, sign * b, -sign * zaxis.x };
	*yaxis = float3{ b, sign + zaxis.y * zaxis.y * a, -zaxis.y };
}
static inline float3 sampleLambertian( const Math& M, float a, float b, const float3& Ng ) // :134-151
{
	float r = sqrtf( a );
	float theta = b * PI * 2.0f;
	float x = r * M.cos_( theta );
	float y = r * M.sin_( theta );
	float z = sqrtf( ss_max( 1.0f - a, 0.0f ) );
	float3 xaxis, yaxis;
	GetOrthonormalBasis( Ng, &xaxis, &yaxis );
	return xaxis * x + yaxis * y + Ng * z;
}
static inline float3 rawReflectance( uchar4 c ) { return { (float)c.x / 255.0f, (float)c.y / 255.0f, (float)c.z / 255.0f }; } // :160-166
template <class T> static inline float luminance( T c ) { return 0.2126f * c.x + 0.7152f * c.y + 0.0722f * c.z; }				 // :168-172
static inline float2 getSpherical( const Math& M, float3 n )																		 // :175-180
{
	float phi = M.atan2_( n.z, n.x ) + PI;
	float theta = M.atan2_( sqrtf( n.x * n.x + n.z * n.z ), n.y );
	return { phi / ( PI * 2.0f ), theta / PI };
}
template <class F> static inline int upper_bound_f( F f, int n, float b ) // :182-202
{
	int i = 0, j = n;
	while( i < j )
	{
		const int m = ( i + j ) / 2;
		const float value = f( m );
		if( value <= b ) i = m + 1;
		else j = m;
	}
	return i;
}
ORC_API void orc_sample_lambertian( int mathMode, float a, float b, const float* N3, float* out3 )
{
	Math M = { mathMode };
	float3 r = sampleLambertian( M, a, b, float3{ N3[0], N3[1], N3[2] } );
	out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

// ------------------------------------------------------------------------------------
// HDRI (renderCommon.hpp:204-481; SAT kernels voxKernel.cu:485-608)
// ------------------------------------------------------------------------------------
struct HDRI
{
	std::vector<float4> pixels, pixelsPrimary;
	std::vector<uint32_t> sat;		// uniform
	std::vector<uint32_t> sats[6];	// +x,-x,+y,-y,+z,-z
	int m_width = 0, m_height = 0, m_widthPrimary = 0, m_heightPrimary = 0;
	float m_scale = 1.75f; // :480
	Math M = { 0 };

	// HDRIstoreImportance, voxKernel.cu:485-524
	void storeImportance( std::vector<double>& satF64, int cosWeighted, float3 axis ) const
	{
		for( int pixelY = 0; pixelY < m_height; pixelY++ )
			for( int pixelX = 0; pixelX < m_width; pixelX++ )
			{
				uint32_t pixelIdx = pixelY * m_width + pixelX;
				float dTheta = PI / (float)m_height;
				float dPhi = 2.0f * PI / (float)m_width;
				float theta = pixelY * dTheta;
				float dH = 2.0f * M.sin_( dTheta * 0.5f ) * M.sin_( dTheta * 0.5f + theta );
				float dW = dPhi;
				float sr = dH * dW;
				float4 color = pixels[pixelIdx];
				float w = 1.0f;
				if( cosWeighted )
				{
					float sY = mix( M.cos_( theta ), M.cos_( theta + dTheta ), 0.5f );
					float phi = dPhi * ( (float)pixelX + 0.5f ) + PI;
					float sX = M.cos_( phi );
					float sZ = M.sin_( phi );
					float sinTheta = sqrtf( ss_max( 1.0f - sY * sY, 0.0f ) );
					float3 dirCenter = { sX * sinTheta, sY, sZ * sinTheta };
					w = ss_max( dot( axis, dirCenter ), 0.0f );
				}
				satF64[pixelIdx] = luminance( color ) * sr * w;
			}
	}
	// prefixSumInclusive<double,512>, voxKernel.cu:526-553: Hillis-Steele in "LDS"
	static double prefixSumInclusive512( double prefix, double* sMemIO )
	{
		double tmp[512];
		for( uint32_t offset = 1; offset < 512; offset <<= 1 )
		{
			for( uint32_t t = 0; t < 512; t++ )
			{
				double x = sMemIO[t];
				if( offset <= t ) x += sMemIO[t - offset];
				tmp[t] = x;
			}
			memcpy( sMemIO, tmp, sizeof( tmp ) );
		}
		double sum = sMemIO[511];
		for( int t = 0; t < 512; t++ ) sMemIO[t] += prefix;
		return sum;
	}
	// buildSATh + buildSATv + buildSAT2u32, voxKernel.cu:557-608
	void buildSAT( std::vector<double>& s, std::vector<uint32_t>& out ) const
	{
		double mem[512];
		for( int Y = 0; Y < m_height; Y++ )
		{
			double prefix = 0.0;
			for( int i = 0; i < m_width; i += 512 )
			{
				for( int t = 0; t < 512; t++ )
				{
					int X = i + t;
					mem[t] = X < m_width ? s[Y * m_width + X] : 0.0;
				}
				prefix += prefixSumInclusive512( prefix, mem );
				for( int t = 0; t < 512; t++ )
				{
					int X = i + t;
					if( X < m_width ) s[Y * m_width + X] = mem[t];
				}
			}
		}
		for( int X = 0; X < m_width; X++ )
		{
			double prefix = 0.0;
			for( int i = 0; i < m_height; i += 512 )
			{
				for( int t = 0; t < 512; t++ )
				{
					int Y = i + t;
					mem[t] = Y < m_height ? s[Y * m_width + X] : 0.0;
				}
				prefix += prefixSumInclusive512( prefix, mem );
				for( int t = 0; t < 512; t++ )
				{
					int Y = i + t;
					if( Y < m_height ) s[Y * m_width + X] = mem[t];
				}
			}
		}
		int n = m_width * m_height;
		out.resize( n );
		double sum = s[n - 1];
		for( int i = 0; i < n; i++ ) out[i] = (uint32_t)( s[i] / ( sum ) * (double)0xFFFFFFFFu );
	}
	// HDRI::load, renderCommon.hpp:214-314
	void load( const float* rgba, int width, int height )
	{
		m_width = width;
		m_height = height;
		pixels.resize( (size_t)width * height );
		memcpy( pixels.data(), rgba, sizeof( float4 ) * pixels.size() );
		std::vector<double> satF64( (size_t)width * height );
		storeImportance( satF64, 0, float3{ 0, 0, 0 } );
		buildSAT( satF64, sat );
		const float3 axisList[6] = { { +1, 0, 0 }, { -1, 0, 0 }, { 0, +1, 0 }, { 0, -1, 0 }, { 0, 0, +1 }, { 0, 0, -1 } };
		for( int i = 0; i < 6; i++ )
		{
			storeImportance( satF64, 1, axisList[i] );
			buildSAT( satF64, sats[i] );
		}
	}
	void loadPrimary( const float* rgba, int width, int height ) // :315-326
	{
		m_widthPrimary = width;
		m_heightPrimary = height;
		pixelsPrimary.resize( (size_t)width * height );
		memcpy( pixelsPrimary.data(), rgba, sizeof( float4 ) * pixelsPrimary.size() );
	}
	float3 sampleNearest( float3 direction, bool isPrimary ) const // :354-365
	{
		int w = isPrimary ? m_widthPrimary : m_width;
		int h = isPrimary ? m_heightPrimary : m_height;
		float2 uv = getSpherical( M, direction );
		int x = (int)ss_clamp( uv.x * w, 0.0f, (float)( w - 1.0f ) );
		int y = (int)ss_clamp( uv.y * h, 0.0f, (float)( h - 1.0f ) );
		uint64_t index = (uint64_t)y * w + x;
		float4 c = ( isPrimary && !pixelsPrimary.empty() ) ? pixelsPrimary[index] : pixels[index];
		return float3{ c.x, c.y, c.z } * m_scale;
	}
	uint32_t getPrefixSumExclusiveH( const uint32_t* s, uint32_t x ) const // :437-444
	{
		if( x <= 0 ) return 0;
		return s[m_width * ( m_height - 1 ) + x - 1];
	}
	uint32_t getPrefixSumExclusiveV( const uint32_t* s, uint32_t x, uint32_t y ) const // :445-455
	{
		if( y <= 0 ) return 0;
		uint32_t s0 = x <= 0 ? 0 : s[m_width * ( y - 1 ) + ( x - 1 )];
		uint32_t s1 = s[m_width * ( y - 1 ) + x];
		return s1 - s0;
	}
	uint32_t getCount( const uint32_t* s, uint32_t x, uint32_t y ) const // :456-465
	{
		uint32_t a = ( x <= 0 || y <= 0 ) ? 0 : s[m_width * ( y - 1 ) + ( x - 1 )];
		uint32_t b = ( y <= 0 ) ? 0 : s[m_width * ( y - 1 ) + x];
		uint32_t c = ( x <= 0 ) ? 0 : s[m_width * y + ( x - 1 )];
		uint32_t d = s[m_width * y + x];
		return ( d - b ) + ( a - c );
	}
	bool isEnabled() const { return 0.0f < m_scale; } // :467-470
	// :367-435
	void importanceSample( float3* direction, float3* L, float* srPDF, float3 N, bool axisAligned, float u0, float u1, float u2, float u3 ) const
	{
		const uint32_t* s = sat.data();
		if( axisAligned )
		{
			const float k = 0.8f;
			if( k < N.x ) s = sats[0].data();
			else if( N.x < -k ) s = sats[1].data();
			else if( k < N.y ) s = sats[2].data();
			else if( N.y < -k ) s = sats[3].data();
			else if( k < N.z ) s = sats[4].data();
			else if( N.z < -k ) s = sats[5].data();
		}
		uint32_t X = upper_bound_f( [this, s]( int i ) { return (float)getPrefixSumExclusiveH( s, i ) / (float)0xFFFFFFFFu; }, m_width, u0 ) - 1;
		uint32_t vol = getPrefixSumExclusiveH( s, X + 1 ) - getPrefixSumExclusiveH( s, X );
		uint32_t Y = upper_bound_f( [this, s, X, vol]( int i ) { return (float)getPrefixSumExclusiveV( s, X, i ) / (float)vol; }, m_height, u1 ) - 1;
		float pSelection = (float)getCount( s, X, Y ) / (float)0xFFFFFFFF;
		float dTheta = PI / (float)m_height;
		float dPhi = 2.0f * PI / (float)m_width;
		float theta = Y * dTheta;
		float dH = 2.0f * M.sin_( dTheta * 0.5f ) * M.sin_( dTheta * 0.5f + theta );
		float dW = dPhi;
		float sr = dH * dW;
		float sY = mix( M.cos_( theta ), M.cos_( theta + dTheta ), u2 );
		float phi = dPhi * ( (float)X + u3 ) + PI;
		float sX = M.cos_( phi );
		float sZ = M.sin_( phi );
		float sinTheta = sqrtf( ss_max( 1.0f - sY * sY, 0.0f ) );
		*direction = { sX * sinTheta, sY, sZ * sinTheta };
		*srPDF = pSelection / sr;
		float4 color = pixels[Y * m_width + X];
		*L = float3{ color.x, color.y, color.z } * m_scale;
	}
};

ORC_API void* orc_hdri_create( const float* rgba, int w, int h, const float* rgbaPrimary, int wp, int hp, int mathMode )
{
	HDRI* H = new HDRI();
	H->M.mode = mathMode;
	H->load( rgba, w, h );
	if( rgbaPrimary ) H->loadPrimary( rgbaPrimary, wp, hp );
	return H;
}
ORC_API void orc_hdri_destroy( void* h ) { delete (HDRI*)h; }
ORC_API void orc_hdri_set_scale( void* h, float s ) { ( (HDRI*)h )->m_scale = s; }
// which: 0 = uniform, 1..6 = +x,-x,+y,-y,+z,-z
ORC_API void orc_hdri_get_sat( void* h, int which, uint32_t* out )
{
	HDRI* H = (HDRI*)h;
	const std::vector<uint32_t>& s = which == 0 ? H->sat : H->sats[which - 1];
	memcpy( out, s.data(), s.size() * 4 );
}
ORC_API void orc_hdri_importance_sample( void* h, const float* N3, int axisAligned, const float* u4, float* dir3, float* L3, float* pdf )
{
	HDRI* H = (HDRI*)h;
	float3 d, L;
	H->importanceSample( &d, &L, pdf, float3{ N3[0], N3[1], N3[2] }, axisAligned != 0, u4[0], u4[1], u4[2], u4[3] );
	dir3[0] = d.x; dir3[1] = d.y; dir3[2] = d.z;
	L3[0] = L.x; L3[1] = L.y; L3[2] = L.z;
}
ORC_API void orc_hdri_sample_nearest( void* h, const float* d3, int isPrimary, float* out3 )
{
	float3 r = ( (HDRI*)h )->sampleNearest( float3{ d3[0], d3[1], d3[2] }, isPrimary != 0 );
	out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}

// Radiance .hdr (RGBE) decode: flat or new-style RLE scanlines; value = c * 2^(E-136), E==0 -> 0,
// alpha = 1 (stb_image convention).  prlib's loader is absent from the reference tree, so this
// boundary is "parity unpinned" (SURVEY.md 8c).  Returns 0 on success.
ORC_API int orc_decode_rgbe( const uint8_t* file, int64_t nBytes, int* wOut, int* hOut, float* rgbaOut, int64_t capacityPixels )
{
	int64_t pos = 0;
	int w = 0, h = 0;
	// header: lines until blank line, then resolution line
	bool blank = false;
	while( pos < nBytes )
	{
		int64_t e = pos;
		while( e < nBytes && file[e] != '\n' ) e++;
		std::string line( (const char*)file + pos, (size_t)( e - pos ) );
		pos = e + 1;
		if( !blank )
		{
			if( line.empty() ) blank = true;
			continue;
		}
		if( sscanf( line.c_str(), "-Y %d +X %d", &h, &w ) != 2 ) return 1;
		break;
	}
	if( w <= 0 || h <= 0 ) return 2;
	*wOut = w;
	*hOut = h;
	if( !rgbaOut ) return 0;
	if( (int64_t)w * h > capacityPixels ) return 3;
	std::vector<uint8_t> scan( (size_t)w * 4 );
	for( int y = 0; y < h; y++ )
	{
		if( pos + 4 <= nBytes && w >= 8 && w < 32768 && file[pos] == 2 && file[pos + 1] == 2 && ( file[pos + 2] & 0x80 ) == 0 &&
			( ( file[pos + 2] << 8 ) | file[pos + 3] ) == w )
		{
			pos += 4;
			for( int c = 0; c < 4; c++ )
			{
				int x = 0;
				while( x < w )
				{
					if( pos >= nBytes ) return 4;
					int count = file[pos++];
					if( count > 128 )
					{
						count -= 128;
						if( pos >= nBytes || x + count > w ) return 4;
						uint8_t v = file[pos++];
						for( int k = 0; k < count; k++ ) scan[( x++ ) * 4 + c] = v;
					}
					else
					{
						if( pos + count > nBytes || x + count > w ) return 4;
						for( int k = 0; k < count; k++ ) scan[( x++ ) * 4 + c] = file[pos++];
					}
				}
			}
		}
		else
		{
			if( pos + (int64_t)w * 4 > nBytes ) return 5;
			memcpy( scan.data(), file + pos, (size_t)w * 4 );
			pos += (int64_t)w * 4;
		}
		for( int x = 0; x < w; x++ )
		{
			const uint8_t* p = &scan[x * 4];
			float* o = rgbaOut + ( (size_t)y * w + x ) * 4;
			if( p[3] )
			{
				float f = ldexpf( 1.0f, (int)p[3] - 136 );
				o[0] = p[0] * f;
				o[1] = p[1] * f;
				o[2] = p[2] * f;
			}
			else
			{
				o[0] = o[1] = o[2] = 0.0f;
			}
			o[3] = 1.0f;
		}
	}
	return 0;
}

// ------------------------------------------------------------------------------------
// render (primary cast), voxKernel.cu:437-483 == voxRT.cpp:307-347
// ------------------------------------------------------------------------------------
ORC_API void orc_render_primary( void* sp, const float* cam15, int W, int H, int showVertexColor, uint8_t* rgbaOut, float* tOut, int32_t* nMajorOut, uint32_t* vIndexOut,
								 uint32_t* descentsOut, int nThreads )
{
	Scene* s = (Scene*)sp;
	CameraPinhole pinhole;
	memcpy( &pinhole, cam15, sizeof( pinhole ) );
	parallelFor( (int64_t)W * H, nThreads, [&]( int64_t b, int64_t e, int ) {
		StackElement stack[32];
		for( int64_t pixelIdx = b; pixelIdx < e; pixelIdx++ )
		{
			uint32_t x = (uint32_t)( pixelIdx % W );
			uint32_t y = (uint32_t)( pixelIdx / W );
			float3 ro, rd;
			pinhole.shoot( &ro, &rd, x, y, 0.5f, 0.5f, W, H );
			float t = MAX_FLOAT;
			int nMajor = -1;
			uint32_t vIndex = 0;
			TraceStats st = { 0, 0 };
			s->intersect( stack, ro, rd, &t, &nMajor, &vIndex, false, &st );
			uchar4 colorOut = { 0, 0, 0, 255 };
			if( t != MAX_FLOAT )
			{
				if( showVertexColor )
				{
					colorOut = s->getVoxelColor( vIndex );
				}
				else
				{
					float3 hitN = getHitN( nMajor, rd );
					float3 color = ( hitN + float3{ 1.0f, 1.0f, 1.0f } ) * 0.5f;
					colorOut = { (uint8_t)( 255 * color.x + 0.5f ), (uint8_t)( 255 * color.y + 0.5f ), (uint8_t)( 255 * color.z + 0.5f ), 255 };
				}
			}
			if( rgbaOut ) memcpy( rgbaOut + pixelIdx * 4, &colorOut, 4 );
			if( tOut ) tOut[pixelIdx] = t;
			if( nMajorOut ) nMajorOut[pixelIdx] = nMajor;
			if( vIndexOut ) vIndexOut[pixelIdx] = vIndex;
			if( descentsOut ) descentsOut[pixelIdx] = (uint32_t)st.descents;
		}
	} );
}

// ------------------------------------------------------------------------------------
// renderPT (voxKernel.cu:610-777), one launch = 16 spp per pixel for `iteration`.
// ------------------------------------------------------------------------------------
struct PTCounters
{
	uint64_t rays;			  // intersect() calls
	uint64_t shadowRays;	  // of which isShadowRay
	uint64_t descents;		  // child fetches, non-shadow rays
	uint64_t shadowDescents;  // child fetches, shadow rays
	uint64_t hits;			  // non-shadow rays that hit (1 attribute fetch each)
	uint64_t samples;
};

// One sample of renderPT's task loop, voxKernel.cu:648-760.  Returns L.
static float3 ptSample( const Scene& scene, const HDRI& hdri, const float* pmj, const CameraPinhole& pinhole, const Math& M, uint32_t pixelIdx, uint32_t spp, int W, int H,
						StackElement* stack, PTCounters* cnt, int* pathHits = nullptr )
{
	uint32_t x = pixelIdx % W;
	uint32_t y = pixelIdx / W;
	MurmurHash32 hash( 0 );
	hash.combine( pixelIdx );
	int dim = 0;
	uint32_t stream = hash.getHash();
#define SAMPLE_2D() pmj_sample2d( pmj, spp, dim++, stream )
#define TRACE( ro_, rd_, t_, nm_, vi_, sh_ )                                   \
	{                                                                          \
		TraceStats st_ = { 0, 0 };                                             \
		scene.intersect( stack, ro_, rd_, t_, nm_, vi_, sh_, &st_ );           \
		cnt->rays++;                                                           \
		if( sh_ ) { cnt->shadowRays++; cnt->shadowDescents += st_.descents; }  \
		else { cnt->descents += st_.descents; if( *( t_ ) != MAX_FLOAT ) cnt->hits++; } \
	}
	float2 cam_u01 = SAMPLE_2D();
	float3 ro, rd;
	float2 lens_u01 = SAMPLE_2D();
	pinhole.shootThinLens( &ro, &rd, x, y, cam_u01.x, cam_u01.y, W, H, lens_u01.x, lens_u01.y );

	float3 T = { 1.0f, 1.0f, 1.0f };
	float3 L = { 0, 0, 0 };
	float t = MAX_FLOAT;
	int nMajor = 0;
	uint32_t vIndex = 0;
	TRACE( ro, rd, &t, &nMajor, &vIndex, false );
	int nPathHits = t != MAX_FLOAT ? 1 : 0; // consecutive hits of the path's own rays (primary, then one per bounce): how long the path stays alive
	if( t == MAX_FLOAT ) // :678-689
	{
		float3 env = hdri.sampleNearest( rd, true );
		L += T * env;
	}
	else
	{
		float3 Le = scene.getVoxelEmission( vIndex, false );
		L += T * Le;
	}
	for( int depth = 0; depth < 8 && t != MAX_FLOAT; depth++ ) // :691
	{
		float3 R = rawReflectance( scene.getVoxelColor( vIndex ) );
		float3 hitN = getHitN( nMajor, rd );
		float3 hitP = ro + rd * t;
		if( hdri.isEnabled() ) // :697-716
		{
			float2 u01 = SAMPLE_2D();
			float2 u23 = SAMPLE_2D();
			float3 dir, emissive;
			float p;
			hdri.importanceSample( &dir, &emissive, &p, hitN, true, u01.x, u01.y, u23.x, u23.y );
			float t2 = MAX_FLOAT;
			int nMajor2 = 0;
			uint32_t vIndex2 = 0;
			TRACE( hitP, dir, &t2, &nMajor2, &vIndex2, true );
			if( t2 == MAX_FLOAT )
			{
				L += T * ( R / PI ) * ss_max( dot( hitN, dir ), 0.0f ) * emissive / p;
			}
		}
		T *= R; // :718
		int nSampleExtraDirect = scene.hasEmission ? 1 : 0; // :721
		for( int k = 0; depth == 0 && k < nSampleExtraDirect; k++ )
		{
			float2 u01 = SAMPLE_2D();
			float3 dir = sampleLambertian( M, u01.x, u01.y, hitN );
			float t2 = MAX_FLOAT;
			int nMajor2 = 0;
			uint32_t vIndex2 = 0;
			TRACE( hitP, dir, &t2, &nMajor2, &vIndex2, false );
			float3 Le = scene.getVoxelEmission( vIndex2, true );
			if( t2 != MAX_FLOAT )
			{
				L += T * Le / (float)( 1 + nSampleExtraDirect );
			}
		}
		float2 u01 = SAMPLE_2D(); // :741
		float3 dir = sampleLambertian( M, u01.x, u01.y, hitN );
		ro = hitP;
		rd = dir;
		t = MAX_FLOAT;
		TRACE( ro, rd, &t, &nMajor, &vIndex, false );
		if( t != MAX_FLOAT )
		{
			nPathHits++;
			float3 Le = scene.getVoxelEmission( vIndex, true );
			L += T * Le * ( depth == 0 ? 1.0f / (float)( 1 + nSampleExtraDirect ) : 1.0f ); // :755
		}
	}
	if( pathHits ) *pathHits = nPathHits;
#undef SAMPLE_2D
#undef TRACE
	cnt->samples++;
	return L;
}

// fb: W*H float4, accumulated in place: xyz += sum over the 16 spp IN ASCENDING spp ORDER (the
// reference's LDS atomicAdd order is nondeterministic, voxKernel.cu:763-765; ascending is the
// order the HIP path fixes), w += 16 (:771-774).  samplePathHits (optional): how many of the path's own rays hit -- the wavefront
// implementation keeps a path alive through exactly that many stages, so its stable compaction can be checked index by index.  pixelBegin/pixelEnd restrict the pixel range
// (bounded CPU-baseline samples).  sampleL (optional): per-sample radiance, (pixel*16+s)*3.
ORC_API void orc_render_pt( void* sp, void* hp, const float* pmjTable, const float* cam15, int W, int H, int iteration, int mathMode, float* fb, int64_t pixelBegin,
							int64_t pixelEnd, float* sampleL, uint64_t* counters6, int nThreads, uint8_t* samplePathHits /* optional, (pixel*16+s): 0..9 */ )
{
	const Scene& scene = *(Scene*)sp;
	HDRI& hdri = *(HDRI*)hp;
	hdri.M.mode = mathMode;
	Math M = { mathMode };
	CameraPinhole pinhole;
	memcpy( &pinhole, cam15, sizeof( pinhole ) );
	const int nBatchSpp = 16;
	if( pixelEnd < 0 ) pixelEnd = (int64_t)W * H;
	std::vector<PTCounters> cnts( std::max( nThreads, 1 ) );
	memset( cnts.data(), 0, sizeof( PTCounters ) * cnts.size() );
	parallelFor( pixelEnd - pixelBegin, nThreads, [&]( int64_t b, int64_t e, int k ) {
		StackElement stack[37]; // PathTracer.hpp:46
		for( int64_t p = pixelBegin + b; p < pixelBegin + e; p++ )
		{
			float3 acc = { 0, 0, 0 };
			for( int s = 0; s < nBatchSpp; s++ )
			{
				uint32_t spp = iteration * nBatchSpp + s; // :642
				int ph = 0;
				float3 L = ptSample( scene, hdri, pmjTable, pinhole, M, (uint32_t)p, spp, W, H, stack, &cnts[k], &ph );
				if( samplePathHits ) samplePathHits[( p - pixelBegin ) * nBatchSpp + s] = (uint8_t)ph;
				acc.x += L.x;
				acc.y += L.y;
				acc.z += L.z;
				if( sampleL )
				{
					float* o = sampleL + ( ( p - pixelBegin ) * nBatchSpp + s ) * 3;
					o[0] = L.x; o[1] = L.y; o[2] = L.z;
				}
			}
			float* f = fb + p * 4;
			f[0] += acc.x;
			f[1] += acc.y;
			f[2] += acc.z;
			f[3] += (float)nBatchSpp;
		}
	} );
	if( counters6 )
	{
		memset( counters6, 0, 6 * 8 );
		for( auto& c : cnts )
		{
			counters6[0] += c.rays;
			counters6[1] += c.shadowRays;
			counters6[2] += c.descents;
			counters6[3] += c.shadowDescents;
			counters6[4] += c.hits;
			counters6[5] += c.samples;
		}
	}
}

// renderResolve, voxKernel.cu:779-795
ORC_API void orc_resolve( const float* fb, int64_t n, int mathMode, uint8_t* rgbaOut )
{
	Math M = { mathMode };
	for( int64_t i = 0; i < n; i++ )
	{
		const float* v = fb + i * 4;
		int r = (int)( 255 * M.pow_( v[0] / v[3], 1.0f / 2.2f ) + 0.5f );
		int g = (int)( 255 * M.pow_( v[1] / v[3], 1.0f / 2.2f ) + 0.5f );
		int b = (int)( 255 * M.pow_( v[2] / v[3], 1.0f / 2.2f ) + 0.5f );
		rgbaOut[i * 4 + 0] = (uint8_t)std::min( r, 255 );
		rgbaOut[i * 4 + 1] = (uint8_t)std::min( g, 255 );
		rgbaOut[i * 4 + 2] = (uint8_t)std::min( b, 255 );
		rgbaOut[i * 4 + 3] = 255;
	}
}

// Stable stream compaction semantics (StreamCompaction.hpp:87-184): dst index of a kept item =
// number of kept items with a smaller source index.  Returns the kept count.
ORC_API int64_t orc_compact_indices( const uint8_t* keep, int64_t n, uint32_t* dstIndexOut /* n, 0xFFFFFFFF for dropped */, uint32_t* srcOfDstOut /* kept */ )
{
	int64_t k = 0;
	for( int64_t i = 0; i < n; i++ )
	{
		if( keep[i] )
		{
			if( dstIndexOut ) dstIndexOut[i] = (uint32_t)k;
			if( srcOfDstOut ) srcOfDstOut[k] = (uint32_t)i;
			k++;
		}
		else if( dstIndexOut )
		{
			dstIndexOut[i] = 0xFFFFFFFFu;
		}
	}
	return k;
}

// detmath probes (for tests/test_detmath.py)
ORC_API void orc_detmath_eval( int which, const float* x, const float* y, int64_t n, float* out )
{
	for( int64_t i = 0; i < n; i++ )
	{
		switch( which )
		{
		case 0: out[i] = mvrt_sin( x[i] ); break;
		case 1: out[i] = mvrt_cos( x[i] ); break;
		case 2: out[i] = mvrt_atan2( y[i], x[i] ); break;
		case 3: out[i] = mvrt_pow( x[i], y[i] ); break;
		case 4: out[i] = mvrt_log( x[i] ); break;
		case 5: out[i] = mvrt_exp( x[i] ); break;
		}
	}
}

ORC_API int orc_struct_sizes( int* out ) // SURVEY Appendix A struct sizes
{
	out[0] = (int)sizeof( OctreeNode );
	out[1] = (int)sizeof( StackElement );
	out[2] = (int)sizeof( OctreeTask );
	out[3] = (int)sizeof( VoxelAttirb );
	out[4] = (int)sizeof( CameraPinhole );
	return 5;
}
