// mvrt_common.h -- POD types and small host/device helpers shared by every translation unit of
// libmvrt_hip.so.  gfx950 only (wave64).  All fp32 arithmetic is compiled with -ffp-contract=off so
// that every traversal decision equals the CPU oracle's bit for bit (DESIGN.md "FP rules").
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvrt_detmath.h"

#define MVRT_HDI __host__ __device__ __forceinline__
#define MVRT_DI __device__ __forceinline__

#define MVRT_MAXF 3.402823466e+38F // reference MAX_FLOAT, vectorMath.hpp:79
#define MVRT_LEAF 0xFFFFFFFFu	   // child pointer of a voxel / absent child, voxKernel.cu:254,299
#define MVRT_SPP_PER_STEP 16	   // nBatchSpp, voxKernel.cu:631
#define MVRT_MAX_DEPTH 8		   // bounce limit, voxKernel.cu:691
#define MVRT_TILE_PIXELS 256	   // RENDER_NUMBER_OF_THREAD, renderCommon.hpp:13

struct f3
{
	float x, y, z;
};
MVRT_HDI f3 mk3( float x, float y, float z ) { return f3{ x, y, z }; }
MVRT_HDI f3 operator+( f3 a, f3 b ) { return f3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
MVRT_HDI f3 operator-( f3 a, f3 b ) { return f3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
MVRT_HDI f3 operator*( f3 a, f3 b ) { return f3{ a.x * b.x, a.y * b.y, a.z * b.z }; }
MVRT_HDI f3 operator*( f3 a, float b ) { return f3{ a.x * b, a.y * b, a.z * b }; }
MVRT_HDI f3 operator/( f3 a, float b ) { return f3{ a.x / b, a.y / b, a.z / b }; }
MVRT_HDI float dot3( f3 a, f3 b ) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// the reference's ss_max / ss_min are plain ternaries (vectorMath.hpp:100-108); keep their NaN behaviour
MVRT_HDI float smax( float x, float y ) { return ( x < y ) ? y : x; }
MVRT_HDI float smin( float x, float y ) { return ( y < x ) ? y : x; }
MVRT_HDI float sabs( float x ) { return x >= 0.0f ? x : -x; }
MVRT_HDI float max3f( float a, float b, float c ) { return smax( smax( a, b ), c ); }
MVRT_HDI float min3f( float a, float b, float c ) { return smin( smin( a, b ), c ); }
MVRT_HDI float mixf( float a, float b, float t ) { return a + ( b - a ) * t; }

// ---- Murmur3 x86_32 word hashing (voxCommon.hpp:45-108) ------------------------------------------
MVRT_HDI uint32_t rotl32( uint32_t x, int r ) { return ( x << r ) | ( x >> ( 32 - r ) ); }
MVRT_HDI uint32_t fmix32( uint32_t h )
{
	h ^= h >> 16;
	h *= 0x85ebca6bu;
	h ^= h >> 13;
	h *= 0xc2b2ae35u;
	h ^= h >> 16;
	return h;
}
MVRT_HDI uint32_t murmurStep( uint32_t h1, uint32_t k1 )
{
	k1 *= 0xcc9e2d51u;
	k1 = rotl32( k1, 15 );
	k1 *= 0x1b873593u;
	h1 ^= k1;
	h1 = rotl32( h1, 13 );
	return h1 * 5u + 0xe6546b64u;
}
MVRT_HDI uint32_t hashCombine2( uint32_t a, uint32_t b ) { return fmix32( murmurStep( a, b ) ^ 4u ); }
MVRT_HDI uint32_t hashCombine3( uint32_t a, uint32_t b, uint32_t c ) { return fmix32( murmurStep( murmurStep( a, b ), c ) ^ 8u ); }

// ---- 63-bit 3-D Morton codes (voxKernel.cu:11-44) ---------------------------------------------
MVRT_HDI uint64_t splitBy3( uint32_t a )
{
	uint64_t x = a & 0x1FFFFFu;
	x = ( x | x << 32 ) & 0x1f00000000ffffull;
	x = ( x | x << 16 ) & 0x1f0000ff0000ffull;
	x = ( x | x << 8 ) & 0x100f00f00f00f00full;
	x = ( x | x << 4 ) & 0x10c30c30c30c30c3ull;
	x = ( x | x << 2 ) & 0x1249249249249249ull;
	return x;
}
MVRT_HDI uint64_t mortonEncode( uint32_t x, uint32_t y, uint32_t z ) { return splitBy3( x ) | splitBy3( y ) << 1 | splitBy3( z ) << 2; }

// ---- Owen-scrambled PMJ02 lookup (pmjSampler.hpp:62-102,158-173) -----------------------------------
#define MVRT_PMJ_LENGTH 4096
#define MVRT_PMJ_NSEQ 128
MVRT_HDI uint32_t laineKarras( uint32_t x, uint32_t seed )
{
	x += seed;
	x ^= x * 0x6c50b47cu;
	x ^= x * 0xb82f1e52u;
	x ^= x * 0xc7afe638u;
	x ^= x * 0x8d22f6e6u;
	return x;
}
MVRT_HDI uint32_t reverseBits32( uint32_t v )
{
#if defined( __HIP_DEVICE_COMPILE__ )
	return __brev( v );
#else
	v = ( ( v >> 1 ) & 0x55555555u ) | ( ( v & 0x55555555u ) << 1 );
	v = ( ( v >> 2 ) & 0x33333333u ) | ( ( v & 0x33333333u ) << 2 );
	v = ( ( v >> 4 ) & 0x0F0F0F0Fu ) | ( ( v & 0x0F0F0F0Fu ) << 4 );
	v = ( ( v >> 8 ) & 0x00FF00FFu ) | ( ( v & 0x00FF00FFu ) << 8 );
	return ( v >> 16 ) | ( v << 16 );
#endif
}
MVRT_HDI uint32_t nestedUniformScramble( uint32_t x, uint32_t seed ) { return reverseBits32( laineKarras( reverseBits32( x ), seed ) ); }
MVRT_HDI float scrambleF32( float x, uint32_t seed )
{
	x += 1.0f;
	uint32_t bits = mvrt_f2u( x );
	uint32_t s = 0x3f800000u | ( nestedUniformScramble( bits & 0x7FFFFFu, seed ) & 0x7FFFFFu );
	return mvrt_u2f( s ) - 1.0f;
}
struct f2
{
	float x, y;
};
MVRT_DI f2 pmjSample2d( const float2* __restrict__ table, uint32_t sampleIdx, uint32_t dimension, uint32_t stream )
{
	sampleIdx = nestedUniformScramble( sampleIdx, hashCombine3( stream, dimension, 31082745u ) ) & ( MVRT_PMJ_LENGTH - 1 );
	dimension = nestedUniformScramble( dimension, hashCombine2( stream, 54761983u ) ) & ( MVRT_PMJ_NSEQ - 1 );
	float2 v = table[dimension * MVRT_PMJ_LENGTH + sampleIdx];
	f2 r;
	r.x = scrambleF32( v.x, hashCombine3( stream, dimension, 83927105u ) );
	r.y = scrambleF32( v.y, hashCombine3( stream, dimension, 12654890u ) );
	return r;
}

// ---- camera (renderCommon.hpp:17-84) ---------------------------------------------------------------
struct CameraPinhole
{
	f3 o, front, up, right;
	float tanHthetaY, lensR, focus;
};
static_assert( sizeof( CameraPinhole ) == 60, "CameraPinhole must match the reference's 60-byte layout" );

MVRT_HDI void cameraShoot( const CameraPinhole& c, f3* ro, f3* rd, int x, int y, float xo, float yo, int W, int H ) // :37-49
{
	float xf = ( x + xo ) / W;
	float yf = ( y + yo ) / H;
	f3 d = c.right * mixf( -c.tanHthetaY, c.tanHthetaY, xf ) * (float)W / (float)H + c.up * mixf( c.tanHthetaY, -c.tanHthetaY, yf ) + c.front;
	*ro = c.o;
	*rd = d;
}
MVRT_HDI void cameraShootThinLens( const CameraPinhole& c, f3* ro, f3* rd, int x, int y, float xo, float yo, int W, int H, float u0, float u1 ) // :50-75
{
	float xf = ( x + xo ) / W;
	float yf = ( y + yo ) / H;
	f3 focalP = mk3( c.focus * mixf( -c.tanHthetaY, c.tanHthetaY, xf ) * (float)W / (float)H, c.focus * mixf( c.tanHthetaY, -c.tanHthetaY, yf ), c.focus );
	f3 lensP = mk3( mixf( -c.lensR, c.lensR, u0 ), mixf( -c.lensR, c.lensR, u1 ), 0.0f );
	f3 dir = focalP - lensP;
	*rd = c.right * dir.x + c.up * dir.y + c.front * dir.z;
	*ro = c.o + c.right * lensP.x + c.up * lensP.y + c.front * lensP.z;
}

// ---- scene on the device ---------------------------------------------------------------------------
// One node = one 64-byte line: children[8] then nVoxelsPSum[8] (the reference's 68-byte AoS node minus its
// leading mask byte; the mask lives in the parent's pointer bits 24-31, or in `masks` for the
// non-embedded variant).  Same indices as the reference layout, so results are unchanged.
struct alignas( 64 ) Node64
{
	uint32_t children[8];
	uint32_t psum[8]; // embedded flavour: nVoxelsPSum.  Non-embedded flavour: psum[0..1] = the 8 CHILD masks (one byte each),
					  // so a descent reads pointer and mask of the child from ONE line; nVoxelsPSum then lives in SvoDev::psumCold
};
struct SvoDev
{
	const Node64* nodes;
	const uint8_t* masks; // per-node own mask (used for the root and by the non-embedded variant)
	const uint32_t* psumCold; // non-embedded flavour only: nVoxelsPSum[node * 8 + child] (read by voxelIndexFromPath only)
	const uint2* attrs;	  // {color rgba8, emission rgba8}
	uint32_t nNodes, nVoxels;
	f3 lower, upper;
	float dps, emissionScale;
	uint32_t hasEmission;
	uint32_t embedded;
	uint32_t levels;
	uint32_t rootIndex; // nNodes - 1
	uint32_t rootMask;
	// embedded flavour: what the first topLevels levels of the nVoxelsPSum walk add up to, per path prefix: topTable[prefix] = { node reached
	// (index), sum of nVoxelsPSum along the prefix }.  The top of a DAG is a tree (sharing starts near the leaves), the table is per PATH, so it
	// is exact either way; entries of prefixes that leave the octree are never looked up (only paths of real hits are resolved).
	const uint32_t* kids; // embedded flavour: children[8] per node, 32 B per node -- what the traversal reads (nodes[] keeps the full lines for the psum walk)
	const uint2* topTable;
	uint32_t topLevels; // 0 = no table
	// tree flavour (svo_build.hip): `nodes` are two-level bricks, treeRoot is where the traversal starts (a brick index, or a voxel index for a
	// one-level octree); hits report the voxel index directly
	uint32_t tree, treeRoot;
	// octrees built here (not the tree flavour): the CELL INDEX (voxelIndexFromPath, traverse_stream.h).  A cell = a parent of voxels (last-but-one level), a block =
	// the 8 x 8 x 8 cells below one node of the fourth level above the voxels.  cellBlocks[block code] = number of the block's dense array of 512 entries
	// { Morton rank of the cell's first voxel, mask of its voxels } in cellEntries (~0 = no voxel in the block); null = walk nVoxelsPSum
	const uint32_t* cellBlocks;
	const uint2* cellEntries;
	uint32_t cellBits; // bits of a cell code that index inside a block: 9, fewer for octrees of fewer than 4 levels
	uint32_t leafPsumIsPopcount; // embedded flavour: nVoxelsPSum of the parents of voxels = exclusive popcount of their mask (every octree this library builds; checked for uploads)
};

MVRT_DI f3 getHitN( int major, f3 rd ) // voxCommon.hpp:564-577
{
	f3 n = mk3( 0.0f, 0.0f, 0.0f );
	if( major == 0 ) n.z = 0.0f < rd.z ? -1.0f : 1.0f;
	else if( major == 1 ) n.x = 0.0f < rd.x ? -1.0f : 1.0f;
	else if( major == 2 ) n.y = 0.0f < rd.y ? -1.0f : 1.0f;
	return n;
}
MVRT_DI f3 rawReflectance( uint32_t rgba8 ) // renderCommon.hpp:160-166
{
	return mk3( (float)( rgba8 & 0xFF ) / 255.0f, (float)( ( rgba8 >> 8 ) & 0xFF ) / 255.0f, (float)( ( rgba8 >> 16 ) & 0xFF ) / 255.0f );
}

// ---- HDRI on the device (renderCommon.hpp:204-481) -------------------------------------------------
struct HdriDev
{
	const float4* pixels;
	const float4* pixelsPrimary;
	const uint32_t* sat;	 // uniform
	const uint32_t* sats[6]; // +x,-x,+y,-y,+z,-z
	int width, height, widthPrimary, heightPrimary;
	float scale;
};

// error plumbing -------------------------------------------------------------------------------------
void mvrtSetError( const char* fmt, ... );
#define MVRT_HIP( expr )                                                                              \
	do                                                                                                \
	{                                                                                                 \
		hipError_t e_ = ( expr );                                                                     \
		if( e_ != hipSuccess )                                                                        \
		{                                                                                             \
			mvrtSetError( "%s failed: %s (%s:%d)", #expr, hipGetErrorString( e_ ), __FILE__, __LINE__ ); \
			return 1;                                                                                 \
		}                                                                                             \
	} while( 0 )

// Experiment knobs: the product has none -- every tuning value is the constant its measurement settled (profiles/*_experiments.txt).  A build with
// -DMVRT_EXPERIMENT (tools/build_variant.sh) reads the same values from the environment for A/B runs.
#ifdef MVRT_EXPERIMENT
#include <stdlib.h>
static inline long long mvrtKnob( const char* name, long long settled ) { const char* e = getenv( name ); return e ? atoll( e ) : settled; }
#else
static inline long long mvrtKnob( const char*, long long settled ) { return settled; }
#endif

static inline uint32_t divUp( uint64_t a, uint64_t b ) { return (uint32_t)( ( a + b - 1 ) / b ); }
