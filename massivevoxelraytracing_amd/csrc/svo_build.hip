// svo_build.hip -- IntersectorOctreeGPU::build on the GPU (reference IntersectorOctreeGPU.hpp:40-241 with
// kernels voxKernel.cu:58-435), redesigned so that the result is DETERMINISTIC and numbered exactly
// like the CPU reference builder buildOctreeDAGReference (IntersectorOctree.hpp:11-123):
//
//   voxelize (six-separating, Schwarz-Seidel; voxelization.hpp:124-337)   [kVoxCount, kVoxelize]
//   radix sort (morton, attribute) pairs                                   [rocPRIM via hipCUB; replaces tinyhipradixsort]
//   unique with integer-mean attributes (voxKernel.cu:170-243)             [head flags -> stable compaction -> kUniqueEmit]
//   one octree level per pass, bottom-up (voxKernel.cu:269-424):
//       group <=8 siblings -> candidate node {mask, children[8], nVoxelsPSum[8]}
//       DAG de-duplication WITHOUT the reference's racy CAS hash table: 64-bit content hash,
//       stable radix sort of (hash, group), equal-content runs collapse onto their FIRST group,
//       new node index = rank of that first occurrence in group order = creation order of the
//       CPU reference (std::map find-or-insert while walking groups in morton order).
//   embed child masks into child pointers (voxKernel.cu:426-435)
//
// All fp32 arithmetic of the voxelizer is contraction-free and matches the oracle bit for bit.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "launch.h"

#define WAVE 64
#define BB 256

namespace
{
struct Buf
{
	void* p = nullptr;
	int alloc( uint64_t bytes )
	{
		release();
		MVRT_HIP( hipMalloc( &p, bytes ? bytes : 1 ) );
		return 0;
	}
	void release()
	{
		if( p ) (void)hipFree( p );
		p = nullptr;
	}
	void* detach()
	{
		void* r = p;
		p = nullptr;
		return r;
	}
	~Buf() { release(); }
	template <class T> T* as() const { return (T*)p; }
};

// ---------------------------------------------------------------------------------------------------
// voxelizer (voxelization.hpp:124-337), six-separating only -- the GPU reference hard-codes it
// (voxKernel.cu:68,109)
// ---------------------------------------------------------------------------------------------------
struct i2
{
	int x, y;
};
MVRT_DI void proj2( f3 p, int axis, float* a, float* b ) // project2plane, voxelization.hpp:15-34
{
	if( axis == 0 ) { *a = p.x; *b = p.y; }
	else if( axis == 1 ) { *a = p.y; *b = p.z; }
	else { *a = p.z; *b = p.x; }
}
MVRT_DI float projRem( f3 p, int axis ) { return axis == 0 ? p.z : ( axis == 1 ? p.x : p.y ); } // :36-48
MVRT_DI void proj2i( int x, int y, int z, int axis, int* a, int* b )
{
	if( axis == 0 ) { *a = x; *b = y; }
	else if( axis == 1 ) { *a = y; *b = z; }
	else { *a = z; *b = x; }
}
MVRT_DI int projRemI( int x, int y, int z, int axis ) { return axis == 0 ? z : ( axis == 1 ? x : y ); }
MVRT_DI f3 cross3( f3 a, f3 b ) { return mk3( a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x ); }

struct VT
{
	int major;
	int lox, loy, hix, hiy, loz, hiz;
	float dC[3][3], nX[3][3], nY[3][3];
	float ox, oy, oz;
	float kx, ky, cSix, cMax, cMin;
	bool conservative; // voxelization.hpp's sixSeparating == false: every voxel the triangle touches (:186-189, :296-301)

	MVRT_DI void init( f3 v0, f3 v1, f3 v2, f3 origin, float dps, int gridRes, bool conservativeMode = false ) // :148-229
	{
		conservative = conservativeMode;
		f3 e01 = v1 - v0;
		f3 e12 = v2 - v1;
		f3 n = cross3( e01, e12 );
		{
			float x = sabs( n.x ), y = sabs( n.y ), z = sabs( n.z ); // majorAxis, :84-94
			major = ( x < y ) ? ( y < z ? 0 : 2 ) : ( x < z ? 0 : 1 );
		}
		f3 bl = mk3( smin( smin( v0.x, v1.x ), v2.x ), smin( smin( v0.y, v1.y ), v2.y ), smin( smin( v0.z, v1.z ), v2.z ) );
		f3 bu = mk3( smax( smax( v0.x, v1.x ), v2.x ), smax( smax( v0.y, v1.y ), v2.y ), smax( smax( v0.z, v1.z ), v2.z ) );
		int lx = (int)floorf( ( bl.x - origin.x ) / dps ), ly = (int)floorf( ( bl.y - origin.y ) / dps ), lz = (int)floorf( ( bl.z - origin.z ) / dps );
		int ux = (int)floorf( ( bu.x - origin.x ) / dps ), uy = (int)floorf( ( bu.y - origin.y ) / dps ), uz = (int)floorf( ( bu.z - origin.z ) / dps );
		lx = lx < 0 ? 0 : lx;
		ly = ly < 0 ? 0 : ly;
		lz = lz < 0 ? 0 : lz;
		ux = ux > gridRes - 1 ? gridRes - 1 : ux;
		uy = uy > gridRes - 1 ? gridRes - 1 : uy;
		uz = uz > gridRes - 1 ? gridRes - 1 : uz;
		proj2i( lx, ly, lz, major, &lox, &loy );
		proj2i( ux, uy, uz, major, &hix, &hiy );
		loz = projRemI( lx, ly, lz, major );
		hiz = projRemI( ux, uy, uz, major );
#pragma unroll
		for( int axis = 0; axis < 3; axis++ )
		{
			float ax[3], ay[3];
			proj2( v0, axis, &ax[0], &ay[0] );
			proj2( v1, axis, &ax[1], &ay[1] );
			proj2( v2, axis, &ax[2], &ay[2] );
			float rem = projRem( n, axis );
			float nSign = 0.0f < rem ? 1.0f : -1.0f;
#pragma unroll
			for( int edge = 0; edge < 3; edge++ )
			{
				int e1 = ( edge + 1 ) % 3;
				float ex = ax[e1] - ax[edge], ey = ay[e1] - ay[edge];
				float nex = -ey * nSign, ney = ex * nSign;
				nX[axis][edge] = nex;
				nY[axis][edge] = ney;
				if( conservative ) // max(ne.x*dp.x, 0) + max(ne.y*dp.y, 0) - dot(ne, a)   (:186-189)
				{
					dC[axis][edge] = smax( nex * dps, 0.0f ) + smax( ney * dps, 0.0f ) - ( nex * ax[edge] + ney * ay[edge] );
				}
				else // six-separating: dot(ne, dp*0.5 - a) + 0.5*dps*max(|ne.x|,|ne.y|)   (:190-193)
				{
					float hx = dps * 0.5f - ax[edge], hy = dps * 0.5f - ay[edge];
					dC[axis][edge] = ( nex * hx + ney * hy ) + 0.5f * dps * smax( sabs( nex ), sabs( ney ) );
				}
			}
		}
		proj2( origin, major, &ox, &oy );
		oz = projRem( origin, major );
		float v0x, v0y;
		proj2( v0, major, &v0x, &v0y );
		float v0z = projRem( v0, major );
		float nx, ny;
		proj2( n, major, &nx, &ny );
		float nz = projRem( n, major );
		kx = -nx / nz;
		ky = -ny / nz;
		float K = -kx * v0x - ky * v0y + v0z;
		cMax = K + dps * ( smax( kx, 0.0f ) + smax( ky, 0.0f ) ); // :225-227
		cMin = K + dps * ( smin( kx, 0.0f ) + smin( ky, 0.0f ) );
		cSix = K + 0.5f * dps * ( kx + ky );
	}
	MVRT_DI i2 yRange( int x, float dps ) const // :235-274
	{
		float xcoord = ox + x * dps;
		float miny = -3.402823466e+38F, maxy = 3.402823466e+38F;
		for( int edge = 0; edge < 3; edge++ )
		{
			float nex = nX[major][edge], ney = nY[major][edge], d = dC[major][edge];
			if( ney == 0.0f )
			{
				if( -nex * xcoord <= d ) continue;
				return i2{ 1, -1 };
			}
			float k = -( xcoord * nex + d ) / ney;
			if( 0.0f < ney ) miny = smax( miny, k );
			else maxy = smin( maxy, k );
		}
		float minIndexF = smax( ( miny - oy ) / dps, -2147483648.0f );
		float maxIndexF = smin( ( maxy - oy ) / dps, 2147483520.0f );
		int lowerY = (int)ceilf( minIndexF );
		int upperY = (int)floorf( maxIndexF );
		lowerY = lowerY < loy ? loy : lowerY;
		upperY = upperY > hiy ? hiy : upperY;
		return i2{ lowerY, upperY };
	}
	MVRT_DI i2 zRange( int x, int y, float dps ) const // :275-306
	{
		float px = ox + dps * x, py = oy + dps * y;
		float var = kx * px + ky * py;
		int zmin, zmax;
		if( conservative ) // :296-301
		{
			float tmax = var + cMax, tmin = var + cMin;
			zmin = (int)floorf( ( tmin - oz ) / dps );
			zmax = (int)floorf( ( tmax - oz ) / dps );
		}
		else
		{
			float tsix = var + cSix;
			float indexf = ( tsix - oz ) / dps;
			float zf = floorf( indexf );
			int z = (int)zf;
			zmin = indexf == zf ? z - 1 : z;
			zmax = z;
		}
		zmin = zmin < loz ? loz : zmin;
		zmax = zmax > hiz ? hiz : zmax;
		return i2{ zmin, zmax };
	}
	MVRT_DI f3 point( int x, int y, int z, float dps ) const // :307-312 + unProjectPlane :96-108
	{
		float a = ox + dps * x, b = oy + dps * y, r = oz + (float)z * dps;
		return major == 0 ? mk3( a, b, r ) : ( major == 1 ? mk3( r, a, b ) : mk3( b, r, a ) );
	}
	MVRT_DI bool intersect( f3 p ) const // :318-336
	{
		if( loz == hiz ) return true;
		for( int axis = 0; axis < 3; axis++ )
		{
			if( axis == major ) continue;
			float a, b;
			proj2( p, axis, &a, &b );
			for( int edge = 0; edge < 3; edge++ )
			{
				float d = nX[axis][edge] * a + nY[axis][edge] * b + dC[axis][edge];
				if( d < 0.0f ) return false;
			}
		}
		return true;
	}
};

MVRT_DI f3 loadV( const float* p, uint64_t i ) { return mk3( p[i * 3], p[i * 3 + 1], p[i * 3 + 2] ); }

// closestBarycentricCoordinateOnTriangle, vectorMath.hpp:249-284
MVRT_DI f3 closestBary( f3 v0, f3 v1, f3 v2, f3 P )
{
	f3 d0 = v0 - P, d1 = v1 - P, d2 = v2 - P;
	f3 e0 = v2 - v0, e1 = v0 - v1, e2 = v1 - v2;
	f3 Ng = cross3( e2, e0 );
	float U = dot3( cross3( d2, d0 ), Ng );
	float V = dot3( cross3( d0, d1 ), Ng );
	float W = dot3( cross3( d1, d2 ), Ng );
	if( U < 0.0f )
	{
		V = dot3( mk3( -d0.x, -d0.y, -d0.z ), e0 );
		W = dot3( d2, e0 );
	}
	else if( V < 0.0f )
	{
		W = dot3( mk3( -d1.x, -d1.y, -d1.z ), e1 );
		U = dot3( d0, e1 );
	}
	else if( W < 0.0f )
	{
		U = dot3( mk3( -d2.x, -d2.y, -d2.z ), e2 );
		V = dot3( d1, e2 );
	}
	f3 bc = mk3( smax( 0.0f, U ), smax( 0.0f, V ), smax( 0.0f, W ) );
	return bc / ( bc.x + bc.y + bc.z );
}

// voxCount (EMIT = false, voxKernel.cu:58-90) and voxelize (EMIT = true, :91-166) share one body.
// The reference gives every triangle one thread, whatever its size; a wall of two triangles across a 4096^3 grid is then 16.7 M cell tests on ONE lane (seconds,
// with 63 lanes of its wave and the rest of the chip idle).  Here a lane only does its own triangle when its footprint is small; triangles whose projected bounding
// rectangle exceeds VOX_BIG_CELLS cells are done by the whole WAVE afterwards, one after the other, the columns of the footprint dealt to the 64 lanes.  Which lane
// emits which voxel -- and so the order of the dumped list -- differs from the reference's; nothing downstream depends on it: the list is sorted by Morton code and the
// attributes of duplicates are integer sums (voxKernel.cu:194-220).
#define VOX_BIG_CELLS 2048
MVRT_DI uint32_t voxColumns( const VT& c, float dps, int x0, int xStep, bool emit, f3 v0, f3 v1, f3 v2, f3 c0, f3 c1, f3 c2, f3 e0, f3 e1, f3 e2, uint64_t* __restrict__ mortonOut,
							 uint64_t* __restrict__ attrOut, uint64_t dst )
{
	uint32_t k = 0;
	for( int x = x0; x <= c.hix; x += xStep )
	{
		i2 yr = c.yRange( x, dps );
		for( int y = yr.x; y <= yr.y; y++ )
		{
			i2 zr = c.zRange( x, y, dps );
			for( int z = zr.x; z <= zr.y; z++ )
			{
				f3 p = c.point( x, y, z, dps );
				if( !c.intersect( p ) ) continue;
				if( emit )
				{
					int cx, cy, cz; // VTContext::i -> unProjectPlane, :110-122,313-316
					if( c.major == 0 ) { cx = x; cy = y; cz = z; }
					else if( c.major == 1 ) { cx = z; cy = x; cz = y; }
					else { cx = y; cy = z; cz = x; }
					mortonOut[dst + k] = mortonEncode( cx, cy, cz );
					f3 bc = closestBary( v0, v1, v2, p );
					f3 bColor = c1 * bc.x + c2 * bc.y + c0 * bc.z;	 // bc.x * c1 + bc.y * c2 + bc.z * c0, voxKernel.cu:148
					f3 bEmission = e1 * bc.x + e2 * bc.y + e0 * bc.z; // :149
					uint64_t a = (uint64_t)(uint8_t)( bColor.x * 255.0f + 0.5f ) | (uint64_t)(uint8_t)( bColor.y * 255.0f + 0.5f ) << 8 |
								 (uint64_t)(uint8_t)( bColor.z * 255.0f + 0.5f ) << 16 | 255ull << 24 | (uint64_t)(uint8_t)( bEmission.x * 255.0f + 0.5f ) << 32 |
								 (uint64_t)(uint8_t)( bEmission.y * 255.0f + 0.5f ) << 40 | (uint64_t)(uint8_t)( bEmission.z * 255.0f + 0.5f ) << 48 | 255ull << 56;
					attrOut[dst + k] = a;
				}
				k++;
			}
		}
	}
	return k;
}
template <bool EMIT>
__global__ void __launch_bounds__( 128 ) kVoxelize( const float* __restrict__ verts, const float* __restrict__ cols, const float* __restrict__ emis, uint32_t nTri,
													 unsigned long long* counter, f3 origin, float dps, int gridRes, uint64_t* __restrict__ mortonOut,
													 uint64_t* __restrict__ attrOut, int conservative )
{
	const uint32_t iTri = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63u;
	const f3 white = mk3( 1, 1, 1 ), black = mk3( 0, 0, 0 ); // voxUtil.hpp:49-61 defaults
	bool big = false;
	if( iTri < nTri )
	{
		f3 v0 = loadV( verts, (uint64_t)iTri * 3 ), v1 = loadV( verts, (uint64_t)iTri * 3 + 1 ), v2 = loadV( verts, (uint64_t)iTri * 3 + 2 );
		VT c;
		c.init( v0, v1, v2, origin, dps, gridRes, conservative != 0 );
		big = (long long)( c.hix - c.lox + 1 ) * (long long)( c.hiy - c.loy + 1 ) > VOX_BIG_CELLS;
		if( !big )
		{
			const uint32_t nVoxels = voxColumns( c, dps, c.lox, 1, false, v0, v1, v2, white, white, white, black, black, black, nullptr, nullptr, 0 );
			if( !EMIT )
			{
				if( nVoxels ) atomicAdd( counter, (unsigned long long)nVoxels );
			}
			else if( nVoxels )
			{
				const uint64_t dst = atomicAdd( counter, (unsigned long long)nVoxels );
				f3 c0 = white, c1 = white, c2 = white, e0 = black, e1 = black, e2 = black;
				if( cols )
				{
					c0 = loadV( cols, (uint64_t)iTri * 3 );
					c1 = loadV( cols, (uint64_t)iTri * 3 + 1 );
					c2 = loadV( cols, (uint64_t)iTri * 3 + 2 );
				}
				if( emis )
				{
					e0 = loadV( emis, (uint64_t)iTri * 3 );
					e1 = loadV( emis, (uint64_t)iTri * 3 + 1 );
					e2 = loadV( emis, (uint64_t)iTri * 3 + 2 );
				}
				voxColumns( c, dps, c.lox, 1, true, v0, v1, v2, c0, c1, c2, e0, e1, e2, mortonOut, attrOut, dst );
			}
		}
	}
	// ---- the big triangles of this wave, one after the other, by all 64 lanes (every lane of the wave reaches this point) ----
	unsigned long long todo = __ballot( big );
	while( todo )
	{
		const uint32_t src = (uint32_t)__builtin_ctzll( todo );
		todo &= todo - 1ull;
		const uint32_t t = ( blockIdx.x * blockDim.x + ( threadIdx.x & ~63u ) ) + src; // the triangle of lane `src` of this wave
		f3 v0 = loadV( verts, (uint64_t)t * 3 ), v1 = loadV( verts, (uint64_t)t * 3 + 1 ), v2 = loadV( verts, (uint64_t)t * 3 + 2 );
		VT c;
		c.init( v0, v1, v2, origin, dps, gridRes, conservative != 0 );
		const uint32_t mine = voxColumns( c, dps, c.lox + (int)lane, 64, false, v0, v1, v2, white, white, white, black, black, black, nullptr, nullptr, 0 );
		uint32_t incl = mine; // inclusive prefix over the lanes
		for( int o = 1; o < 64; o <<= 1 )
		{
			const uint32_t up = __shfl_up( incl, o, 64 );
			if( (int)lane >= o ) incl += up;
		}
		const uint32_t total = __shfl( incl, 63, 64 );
		if( total == 0 ) continue;
		unsigned long long base = 0;
		if( lane == 0 ) base = atomicAdd( counter, (unsigned long long)total );
		if( !EMIT ) continue;
		base = (unsigned long long)__shfl( (uint32_t)base, 0, 64 ) | ( (unsigned long long)__shfl( (uint32_t)( base >> 32 ), 0, 64 ) << 32 );
		f3 c0 = white, c1 = white, c2 = white, e0 = black, e1 = black, e2 = black;
		if( cols )
		{
			c0 = loadV( cols, (uint64_t)t * 3 );
			c1 = loadV( cols, (uint64_t)t * 3 + 1 );
			c2 = loadV( cols, (uint64_t)t * 3 + 2 );
		}
		if( emis )
		{
			e0 = loadV( emis, (uint64_t)t * 3 );
			e1 = loadV( emis, (uint64_t)t * 3 + 1 );
			e2 = loadV( emis, (uint64_t)t * 3 + 2 );
		}
		voxColumns( c, dps, c.lox + (int)lane, 64, true, v0, v1, v2, c0, c1, c2, e0, e1, e2, mortonOut, attrOut, base + ( incl - mine ) );
	}
}

// ---------------------------------------------------------------------------------------------------
// stable compaction helpers (same scheme as kernels_rt.hip: count per 256 -> scan -> rank)
// ---------------------------------------------------------------------------------------------------
template <class Pred>
MVRT_DI void countBody( Pred pred, uint64_t n, uint32_t* __restrict__ blockCount )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * BB + threadIdx.x;
		bool keep = i < n ? pred( i ) : false;
		unsigned long long b = __ballot( keep );
		if( ( threadIdx.x & ( WAVE - 1 ) ) == 0 ) wc[threadIdx.x / WAVE] = __popcll( b );
		__syncthreads();
		if( threadIdx.x == 0 ) blockCount[vb] = wc[0] + wc[1] + wc[2] + wc[3];
		__syncthreads();
	}
}
MVRT_DI uint32_t rankInBlock( bool keep, uint32_t* wc )
{
	unsigned long long b = __ballot( keep );
	uint32_t lane = threadIdx.x & ( WAVE - 1 ), w = threadIdx.x / WAVE;
	uint32_t inWave = __popcll( b & ( ( 1ull << lane ) - 1ull ) );
	if( lane == 0 ) wc[w] = __popcll( b );
	__syncthreads();
	uint32_t off = 0;
	for( uint32_t k = 0; k < w; k++ ) off += wc[k];
	__syncthreads();
	return off + inWave;
}
__global__ void __launch_bounds__( 1024 ) kScanCounts( uint32_t* __restrict__ blockCount, uint64_t nItems, uint32_t* __restrict__ totalOut )
{
	__shared__ uint32_t part[1024];
	uint32_t nBlocks = (uint32_t)( ( nItems + BB - 1 ) / BB );
	uint32_t per = ( nBlocks + 1023 ) / 1024;
	uint32_t beg = threadIdx.x * per;
	uint32_t end = beg + per < nBlocks ? beg + per : nBlocks;
	uint32_t sum = 0;
	for( uint32_t i = beg; i < end; i++ ) sum += blockCount[i];
	part[threadIdx.x] = sum;
	__syncthreads();
	for( uint32_t off = 1; off < 1024; off <<= 1 )
	{
		uint32_t v = part[threadIdx.x];
		if( threadIdx.x >= off ) v += part[threadIdx.x - off];
		__syncthreads();
		part[threadIdx.x] = v;
		__syncthreads();
	}
	uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
	for( uint32_t i = beg; i < end; i++ )
	{
		uint32_t c = blockCount[i];
		blockCount[i] = run;
		run += c;
	}
	if( threadIdx.x == 1023 ) *totalOut = part[1023];
}

// ---- unique (voxKernel.cu:170-243) ---------------------------------------------------------------------
__global__ void __launch_bounds__( BB ) kUniqueCount( const uint64_t* __restrict__ keys, uint64_t n, uint32_t* __restrict__ blockCount )
{
	countBody( [=]( uint64_t i ) { return i == 0 || keys[i - 1] != keys[i]; }, n, blockCount );
}
__global__ void __launch_bounds__( BB ) kUniqueEmit( const uint64_t* __restrict__ keys, const uint64_t* __restrict__ vals, uint64_t n, const uint32_t* __restrict__ blockOff,
													  uint64_t* __restrict__ keysOut, uint2* __restrict__ attrsOut, uint32_t* __restrict__ hasEmission )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * BB + threadIdx.x;
		bool head = i < n ? ( i == 0 || keys[i - 1] != keys[i] ) : false;
		uint32_t r = rankInBlock( head, wc );
		if( head )
		{
			uint64_t morton = keys[i];
			int s[6] = { 0, 0, 0, 0, 0, 0 };
			int cnt = 0;
			for( uint64_t j = i; j < n && keys[j] == morton; j++ ) // :201-210 integer sums
			{
				uint64_t a = vals[j];
				s[0] += (int)( a & 0xFF );
				s[1] += (int)( ( a >> 8 ) & 0xFF );
				s[2] += (int)( ( a >> 16 ) & 0xFF );
				s[3] += (int)( ( a >> 32 ) & 0xFF );
				s[4] += (int)( ( a >> 40 ) & 0xFF );
				s[5] += (int)( ( a >> 48 ) & 0xFF );
				cnt++;
			}
			uint32_t col = (uint32_t)( s[0] / cnt ) | (uint32_t)( s[1] / cnt ) << 8 | (uint32_t)( s[2] / cnt ) << 16 | 255u << 24;
			uint32_t em = (uint32_t)( s[3] / cnt ) | (uint32_t)( s[4] / cnt ) << 8 | (uint32_t)( s[5] / cnt ) << 16 | 255u << 24;
			uint64_t dst = (uint64_t)blockOff[vb] + r;
			keysOut[dst] = morton;
			attrsOut[dst] = make_uint2( col, em );
			if( em & 0xFFFFFFu ) atomicOr( hasEmission, 1u ); // :225-228
		}
	}
}

// ---- octree levels ---------------------------------------------------------------------------------------
struct Task // OctreeTask, voxCommon.hpp:29-36
{
	uint64_t morton;
	uint32_t child;
	uint32_t count;
};
struct Cand // candidate node of one sibling group
{
	uint32_t children[8];
	uint32_t psum[8];
	uint32_t mask;
	uint32_t total;
};

__global__ void __launch_bounds__( BB ) kInitTasks( const uint64_t* __restrict__ morton, uint32_t n, Task* __restrict__ tasks )
{
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB )
	{
		Task t;
		t.morton = morton[i];
		t.child = MVRT_LEAF; // octreeTaskInit, voxKernel.cu:253-255
		t.count = 1;
		tasks[i] = t;
	}
}
__global__ void __launch_bounds__( BB ) kGroupCount( const Task* __restrict__ tasks, uint64_t n, uint32_t* __restrict__ blockCount )
{
	countBody( [=]( uint64_t i ) { return i == 0 || ( tasks[i - 1].morton >> 3 ) != ( tasks[i].morton >> 3 ); }, n, blockCount );
}

MVRT_HDI uint64_t mix64( uint64_t h, uint64_t k )
{
	k *= 0x87c37b91114253d5ull;
	k = ( k << 31 ) | ( k >> 33 );
	k *= 0x4cf5ad432745937full;
	h ^= k;
	h = ( h << 27 ) | ( h >> 37 );
	return h * 5 + 0x52dce729ull;
}
MVRT_HDI uint64_t fmix64( uint64_t k )
{
	k ^= k >> 33;
	k *= 0xff51afd7ed558ccdull;
	k ^= k >> 33;
	k *= 0xc4ceb9fe1a85ec53ull;
	k ^= k >> 33;
	return k;
}

// one thread per sibling group head: gather <= 8 tasks into a candidate node (voxKernel.cu:293-320)
__global__ void __launch_bounds__( BB ) kMakeCandidates( const Task* __restrict__ tasks, uint64_t n, const uint32_t* __restrict__ blockOff, Cand* __restrict__ cands,
														  uint64_t* __restrict__ hashes, uint32_t* __restrict__ groupIds, uint64_t* __restrict__ parentMorton )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * BB + threadIdx.x;
		bool head = i < n ? ( i == 0 || ( tasks[i - 1].morton >> 3 ) != ( tasks[i].morton >> 3 ) ) : false;
		uint32_t r = rankInBlock( head, wc );
		if( head )
		{
			Cand c;
			for( int j = 0; j < 8; j++ )
			{
				c.children[j] = MVRT_LEAF;
				c.psum[j] = 0;
			}
			c.mask = 0;
			uint64_t parent = tasks[i].morton >> 3;
			for( uint64_t j = i; j < n && ( tasks[j].morton >> 3 ) == parent; j++ )
			{
				Task t = tasks[j];
				uint32_t space = (uint32_t)( t.morton & 7 );
				c.mask |= 1u << space;
				c.children[space] = t.child;
				c.psum[space] = t.count;
			}
			uint32_t run = 0;
			for( int j = 0; j < 8; j++ ) // exclusive prefix sum, :313-320
			{
				uint32_t v = c.psum[j];
				c.psum[j] = run;
				run += v;
			}
			c.total = run;
			uint32_t g = blockOff[vb] + r;
			cands[g] = c;
			uint64_t h = mix64( 0x9e3779b97f4a7c15ull, c.mask );
			for( int j = 0; j < 8; j += 2 ) h = mix64( h, (uint64_t)c.children[j] | (uint64_t)c.children[j + 1] << 32 );
			hashes[g] = fmix64( h );
			groupIds[g] = g;
			parentMorton[g] = parent;
		}
	}
}

MVRT_DI bool sameContent( const Cand& a, const Cand& b ) // OctreeNode::operator==, voxCommon.hpp:147-161
{
	if( a.mask != b.mask ) return false;
	for( int j = 0; j < 8; j++ )
		if( a.children[j] != b.children[j] ) return false;
	return true;
}

// in hash-sorted order: run head = first element whose (hash, content) differs from its predecessor.
// headPos[i] = i for heads, 0 otherwise (an inclusive max-scan then gives each element its run head).
__global__ void __launch_bounds__( BB ) kMarkRuns( const uint64_t* __restrict__ sortedHash, const uint32_t* __restrict__ sortedGroup, const Cand* __restrict__ cands, uint32_t n,
													uint32_t* __restrict__ headPos )
{
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB )
	{
		bool head = i == 0 || sortedHash[i] != sortedHash[i - 1] || !sameContent( cands[sortedGroup[i]], cands[sortedGroup[i - 1]] );
		headPos[i] = head ? (uint32_t)i : 0u;
	}
}
// repOf[group] = first (smallest) group of its equal-content run; the sort is stable and groups entered it
// in ascending order, so the run head carries the smallest group id
__global__ void __launch_bounds__( BB ) kScatterReps( const uint32_t* __restrict__ sortedGroup, const uint32_t* __restrict__ headPosScanned, uint32_t n,
													   uint32_t* __restrict__ repOf )
{
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB )
	{
		repOf[sortedGroup[i]] = sortedGroup[headPosScanned[i]];
	}
}
__global__ void __launch_bounds__( BB ) kIdentityReps( uint32_t n, uint32_t* __restrict__ repOf )
{
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB ) repOf[i] = (uint32_t)i;
}
__global__ void __launch_bounds__( BB ) kFirstCount( const uint32_t* __restrict__ repOf, uint64_t n, uint32_t* __restrict__ blockCount )
{
	countBody( [=]( uint64_t g ) { return repOf[g] == (uint32_t)g; }, n, blockCount );
}
// first occurrences get node index nodeBase + rank (creation order); they also write the node
__global__ void __launch_bounds__( BB ) kEmitNodes( const uint32_t* __restrict__ repOf, const Cand* __restrict__ cands, uint64_t n, const uint32_t* __restrict__ blockOff,
													 uint32_t nodeBase, Node64* __restrict__ nodes, uint8_t* __restrict__ masks, uint32_t* __restrict__ nodeOfGroup )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t g = vb * BB + threadIdx.x;
		bool first = g < n ? repOf[g] == (uint32_t)g : false;
		uint32_t r = rankInBlock( first, wc );
		if( first )
		{
			uint32_t idx = nodeBase + blockOff[vb] + r;
			const Cand& c = cands[g];
			Node64 nd;
			for( int j = 0; j < 8; j++ )
			{
				nd.children[j] = c.children[j];
				nd.psum[j] = c.psum[j];
			}
			nodes[idx] = nd;
			masks[idx] = (uint8_t)c.mask;
			nodeOfGroup[g] = idx;
		}
	}
}
__global__ void __launch_bounds__( BB ) kNextTasks( const uint32_t* __restrict__ repOf, const uint32_t* __restrict__ nodeOfGroup, const Cand* __restrict__ cands,
													 const uint64_t* __restrict__ parentMorton, uint32_t n, Task* __restrict__ tasksOut )
{
	for( uint64_t g = (uint64_t)blockIdx.x * BB + threadIdx.x; g < n; g += (uint64_t)gridDim.x * BB )
	{
		Task t;
		t.morton = parentMorton[g];
		t.child = nodeOfGroup[repOf[g]];
		t.count = cands[g].total;
		tasksOut[g] = t;
	}
}
// embedMasks, voxKernel.cu:426-435 / voxCommon.hpp:183-195
__global__ void __launch_bounds__( BB ) kEmbedMasks( Node64* __restrict__ nodes, const uint8_t* __restrict__ masks, uint32_t nNodes )
{
	uint64_t gid = (uint64_t)blockIdx.x * BB + threadIdx.x;
	uint64_t node = gid / 8;
	uint32_t k = gid % 8;
	if( node >= nNodes ) return;
	uint32_t child = nodes[node].children[k];
	if( child == MVRT_LEAF ) return;
	nodes[node].children[k] = child | (uint32_t)masks[child] << 24;
}

// ---- no-DAG path: every sibling group becomes a node, written straight to its final slot (group index = node index
// within the level; the reference's ENABLE_GPU_DAG-off branch, voxKernel.cu:322-334, numbers them with a racing
// atomicInc -- here the numbering is the deterministic group order) ------------------------------------------------------
__global__ void __launch_bounds__( BB ) kMakeNodesDirect( const Task* __restrict__ tasks, uint64_t n, const uint32_t* __restrict__ blockOff, uint32_t nodeBase,
														   Node64* __restrict__ nodes, uint8_t* __restrict__ masks, Task* __restrict__ tasksOut )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * BB + threadIdx.x;
		bool head = i < n ? ( i == 0 || ( tasks[i - 1].morton >> 3 ) != ( tasks[i].morton >> 3 ) ) : false;
		uint32_t r = rankInBlock( head, wc );
		if( head )
		{
			Node64 nd;
			for( int j = 0; j < 8; j++ )
			{
				nd.children[j] = MVRT_LEAF;
				nd.psum[j] = 0;
			}
			uint32_t mask = 0;
			const uint64_t parent = tasks[i].morton >> 3;
			for( uint64_t j = i; j < n && ( tasks[j].morton >> 3 ) == parent; j++ )
			{
				Task t = tasks[j];
				uint32_t space = (uint32_t)( t.morton & 7 );
				mask |= 1u << space;
				nd.children[space] = t.child;
				nd.psum[space] = t.count;
			}
			uint32_t run = 0;
			for( int j = 0; j < 8; j++ )
			{
				uint32_t v = nd.psum[j];
				nd.psum[j] = run;
				run += v;
			}
			const uint32_t g = blockOff[vb] + r;
			const uint64_t idx = (uint64_t)nodeBase + g;
			nodes[idx] = nd;
			masks[idx] = (uint8_t)mask;
			Task o;
			o.morton = parent;
			o.child = (uint32_t)idx;
			o.count = run;
			tasksOut[g] = o;
		}
	}
}
// ---- "tree" flavour (no DAG and too many nodes for embedded masks, or embedding switched off): the nodes are never materialised as 64-byte
// lines.  In a tree built level by level in morton order the children of a node are CONSECUTIVE nodes of the level below, so a node is
// { mask, index of its first child } (5 bytes; for the parents of voxels: index of their first voxel), and what the traversal reads are
// BRICKS: one 16-byte record per node of every second level { the masks of its 8 children, its own mask, where the children of its first
// child start } (four bricks share a 64-byte line) -- one fetch advances the traversal TWO levels (traverse_stream.h, flavour 2). ------------
__global__ void __launch_bounds__( BB ) kMakeNodesTree( const Task* __restrict__ tasks, uint64_t n, const uint32_t* __restrict__ blockOff, uint32_t nodeBase, int bottomLevel,
														 uint8_t* __restrict__ cMask, uint32_t* __restrict__ cFirst, Task* __restrict__ tasksOut )
{
	__shared__ uint32_t wc[BB / WAVE];
	const uint64_t nBlocks = ( n + BB - 1 ) / BB;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * BB + threadIdx.x;
		bool head = i < n ? ( i == 0 || ( tasks[i - 1].morton >> 3 ) != ( tasks[i].morton >> 3 ) ) : false;
		uint32_t r = rankInBlock( head, wc );
		if( head )
		{
			uint32_t mask = 0, run = 0;
			const uint64_t parent = tasks[i].morton >> 3;
			for( uint64_t j = i; j < n && ( tasks[j].morton >> 3 ) == parent; j++ )
			{
				mask |= 1u << (uint32_t)( tasks[j].morton & 7 );
				run += tasks[j].count;
			}
			const uint32_t g = blockOff[vb] + r;
			const uint64_t idx = (uint64_t)nodeBase + g;
			cMask[idx] = (uint8_t)mask;
			cFirst[idx] = bottomLevel ? (uint32_t)i : tasks[i].child; // first voxel (= its vIndex) / first child node: the rest follow consecutively
			Task o;
			o.morton = parent;
			o.child = (uint32_t)idx;
			o.count = run;
			tasksOut[g] = o;
		}
	}
}
// bricks of one (odd) builder level: node r = levelBase + g has children k = cFirst[r] + j in the level below.  grandBase / grandBrickBase
// translate the children's first-child indices (nodes two levels below) into brick indices; childIsBottom: the children are parents of voxels
// and their cFirst already IS a voxel index.
__global__ void __launch_bounds__( BB ) kMakeBricks( const uint8_t* __restrict__ cMask, const uint32_t* __restrict__ cFirst, uint32_t levelBase, uint32_t count, int childIsBottom,
													  uint32_t grandBase, uint32_t grandBrickBase, uint4* __restrict__ bricksOfLevel )
{
	for( uint64_t g = (uint64_t)blockIdx.x * BB + threadIdx.x; g < count; g += (uint64_t)gridDim.x * BB )
	{
		const uint32_t r = levelBase + (uint32_t)g;
		const uint32_t mask = cMask[r];
		uint32_t k = cFirst[r];
		uint32_t lo = 0, hi = 0, base = 0;
		bool first = true;
		for( int c = 0; c < 8; c++ )
		{
			if( mask >> c & 1u )
			{
				const uint32_t m = cMask[k];
				if( first ) base = childIsBottom ? cFirst[k] : grandBrickBase + ( cFirst[k] - grandBase );
				first = false;
				if( c < 4 ) lo |= m << ( 8 * c );
				else hi |= m << ( 8 * ( c - 4 ) );
				k++;
			}
		}
		// A brick is 16 bytes: the masks of the root's eight children (byte c = child c, 0 = absent), the root's own mask, and where the children
		// of its first existing child start (bricks two levels down, or voxels).  The children of consecutive nodes lie back to back -- every level
		// is numbered in Morton order -- so child c's start is base + popcount( child masks before c ): nothing else of the brick is ever read
		// (traverse_stream.h, treeDescend), and the up to eight bricks below one node share 128 bytes.
		bricksOfLevel[g] = make_uint4( lo, hi, mask, base );
	}
}
// distinct parents per level for ALL levels in one pass (the reference's octreeTaskInit counters, voxKernel.cu:257-265)
__global__ void __launch_bounds__( BB ) kLevelCounts( const uint64_t* __restrict__ morton, uint64_t n, int levels, unsigned long long* __restrict__ counts )
{
	// per-workgroup tallies in LDS, one global atomic per level and workgroup at the end (13 hot addresses took 0.64 s of global
	// atomics for 650 M voxels when every wave added to them directly)
	__shared__ unsigned int tally[32];
	if( threadIdx.x < 32 ) tally[threadIdx.x] = 0u;
	__syncthreads();
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB )
	{
		const uint64_t a = morton[i], b = i ? morton[i - 1] : 0;
		for( int l = 0; l < levels; l++ )
		{
			const bool head = i == 0 || ( a >> ( 3 * ( l + 1 ) ) ) != ( b >> ( 3 * ( l + 1 ) ) );
			const unsigned long long m = __ballot( head );
			if( head && ( threadIdx.x & 63 ) == (uint32_t)__builtin_ctzll( m ) ) atomicAdd( &tally[l], (unsigned int)__popcll( m ) );
		}
	}
	__syncthreads();
	if( threadIdx.x < (uint32_t)levels && tally[threadIdx.x] ) atomicAdd( &counts[threadIdx.x], (unsigned long long)tally[threadIdx.x] );
}
// seeded synthetic voxels: uniformly random cells of the grid with hash-derived colours; ~1/256 of them emissive
MVRT_HDI uint64_t splitmix64( uint64_t x )
{
	x += 0x9E3779B97F4A7C15ull;
	x = ( x ^ ( x >> 30 ) ) * 0xBF58476D1CE4E5B9ull;
	x = ( x ^ ( x >> 27 ) ) * 0x94D049BB133111EBull;
	return x ^ ( x >> 31 );
}
__global__ void __launch_bounds__( BB ) kSyntheticVoxels( uint64_t n, uint64_t seed, uint32_t gridRes, uint64_t* __restrict__ mortonOut, uint64_t* __restrict__ attrOut )
{
	for( uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB )
	{
		const uint64_t h = splitmix64( seed + i );
		const uint64_t c = splitmix64( h );
		const uint32_t x = (uint32_t)( h ) & ( gridRes - 1 ), y = (uint32_t)( h >> 21 ) & ( gridRes - 1 ), z = (uint32_t)( h >> 42 ) & ( gridRes - 1 );
		mortonOut[i] = mortonEncode( x, y, z );
		const uint64_t rgb = ( ( c & 0xFFFFFFull ) | 0x404040ull ); // keep reflectance above 0.25
		const uint64_t emissive = ( ( c >> 56 ) == 0 ) ? rgb : 0ull;
		attrOut[i] = rgb | 255ull << 24 | emissive << 32 | 255ull << 56;
	}
}

struct MaxOp
{
	__host__ __device__ uint32_t operator()( uint32_t a, uint32_t b ) const { return a > b ? a : b; }
};

int gridFor( uint64_t n )
{
	uint64_t b = ( n + BB - 1 ) / BB;
	if( b < 1 ) b = 1;
	return (int)( b > 4096 ? 4096 : b );
}
} // namespace

static int buildFromFragments( Buf& keysA, Buf& valsA, unsigned long long totalDumped, int gridRes, int flags, hipStream_t st, unsigned long long* counter, uint32_t* hasEmission,
							   uint32_t* scalarOut, SvoBuildResult* out );

int svoBuildFromTriangles( const float* vertsHost, const float* colsHost, const float* emisHost, uint64_t nVertices, f3 origin, float dps, int gridRes, int flags, hipStream_t st,
						   SvoBuildResult* out )
{
	const uint32_t nTri = (uint32_t)( nVertices / 3 );
	int levels = 0;
	while( ( 1 << levels ) < gridRes ) levels++;

	Buf dVerts, dCols, dEmis, dCounter;
	if( dVerts.alloc( nVertices * 12 ) || dCounter.alloc( 64 ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( dVerts.p, vertsHost, nVertices * 12, hipMemcpyHostToDevice, st ) );
	if( colsHost )
	{
		if( dCols.alloc( nVertices * 12 ) ) return 1;
		MVRT_HIP( hipMemcpyAsync( dCols.p, colsHost, nVertices * 12, hipMemcpyHostToDevice, st ) );
	}
	if( emisHost )
	{
		if( dEmis.alloc( nVertices * 12 ) ) return 1;
		MVRT_HIP( hipMemcpyAsync( dEmis.p, emisHost, nVertices * 12, hipMemcpyHostToDevice, st ) );
	}
	unsigned long long* counter = dCounter.as<unsigned long long>();
	uint32_t* hasEmission = (uint32_t*)( counter + 1 );
	uint32_t* scalarOut = (uint32_t*)( counter + 2 );
	MVRT_HIP( hipMemsetAsync( dCounter.p, 0, 64, st ) );

	// ---- voxelize: count, allocate, emit (IntersectorOctreeGPU.hpp:81-116) ----
	const uint32_t triGrid = divUp( nTri, 128 );
	hipLaunchKernelGGL( kVoxelize<false>, dim3( triGrid ), dim3( 128 ), 0, st, dVerts.as<float>(), (const float*)nullptr, (const float*)nullptr, nTri, counter, origin, dps,
						gridRes, (uint64_t*)nullptr, (uint64_t*)nullptr, ( flags & 4 ) ? 1 : 0 );
	unsigned long long totalDumped = 0;
	MVRT_HIP( hipMemcpyAsync( &totalDumped, counter, 8, hipMemcpyDeviceToHost, st ) );
	MVRT_HIP( hipStreamSynchronize( st ) );
	if( totalDumped == 0 )
	{
		mvrtSetError( "mvrt_svo_build: the triangles touch no voxel of the grid" );
		return 1;
	}
	if( totalDumped >= 0xFFFFFFFFull )
	{
		mvrtSetError( "mvrt_svo_build: %llu voxel fragments exceed the 32-bit index range of this builder", totalDumped );
		return 1;
	}
	Buf keysA, valsA;
	if( keysA.alloc( totalDumped * 8 ) || valsA.alloc( totalDumped * 8 ) ) return 1;
	MVRT_HIP( hipMemsetAsync( counter, 0, 8, st ) );
	hipLaunchKernelGGL( kVoxelize<true>, dim3( triGrid ), dim3( 128 ), 0, st, dVerts.as<float>(), dCols.as<float>(), dEmis.as<float>(), nTri, counter, origin, dps, gridRes,
						keysA.as<uint64_t>(), valsA.as<uint64_t>(), ( flags & 4 ) ? 1 : 0 );

	dVerts.release();
	dCols.release();
	dEmis.release();
	return buildFromFragments( keysA, valsA, totalDumped, gridRes, flags, st, counter, hasEmission, scalarOut, out );
}

int svoBuildSynthetic( uint64_t nRandomVoxels, uint64_t seed, int gridRes, int flags, hipStream_t st, SvoBuildResult* out )
{
	if( nRandomVoxels == 0 || nRandomVoxels >= 0xFFFFFFFFull )
	{
		mvrtSetError( "mvrt_svo_build_synthetic: voxel count must be in [1, 2^32-2]" );
		return 1;
	}
	Buf dCounter, keysA, valsA;
	if( dCounter.alloc( 64 ) || keysA.alloc( nRandomVoxels * 8 ) || valsA.alloc( nRandomVoxels * 8 ) ) return 1;
	MVRT_HIP( hipMemsetAsync( dCounter.p, 0, 64, st ) );
	unsigned long long* counter = dCounter.as<unsigned long long>();
	hipLaunchKernelGGL( kSyntheticVoxels, dim3( gridFor( nRandomVoxels ) ), dim3( BB ), 0, st, nRandomVoxels, seed, (uint32_t)gridRes, keysA.as<uint64_t>(), valsA.as<uint64_t>() );
	return buildFromFragments( keysA, valsA, nRandomVoxels, gridRes, flags, st, counter, (uint32_t*)( counter + 1 ), (uint32_t*)( counter + 2 ), out );
}

static int buildFromFragments( Buf& keysA, Buf& valsA, unsigned long long totalDumped, int gridRes, int flags, hipStream_t st, unsigned long long* counter, uint32_t* hasEmission,
							   uint32_t* scalarOut, SvoBuildResult* out )
{
	(void)counter;
	int levels = 0;
	while( ( 1 << levels ) < gridRes ) levels++;
	const bool dag = !( flags & 1 );
	Buf keysB, valsB;
	if( keysB.alloc( totalDumped * 8 ) || valsB.alloc( totalDumped * 8 ) ) return 1;
	// ---- sort (IntersectorOctreeGPU.hpp:117-124) ----
	{
		size_t tmpBytes = 0;
		MVRT_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, tmpBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), valsA.as<uint64_t>(), valsB.as<uint64_t>(),
													  (uint64_t)totalDumped, 0, 3 * levels, st ) );
		Buf tmp;
		if( tmp.alloc( tmpBytes ) ) return 1;
		MVRT_HIP( hipcub::DeviceRadixSort::SortPairs( tmp.p, tmpBytes, keysA.as<uint64_t>(), keysB.as<uint64_t>(), valsA.as<uint64_t>(), valsB.as<uint64_t>(),
													  (uint64_t)totalDumped, 0, 3 * levels, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );
	}
	keysA.release();
	valsA.release();

	// ---- unique with integer-mean attributes (IntersectorOctreeGPU.hpp:126-137) ----
	Buf blockCnt;
	if( blockCnt.alloc( ( totalDumped / BB + 2 ) * 4 ) ) return 1;
	hipLaunchKernelGGL( kUniqueCount, dim3( gridFor( totalDumped ) ), dim3( BB ), 0, st, keysB.as<uint64_t>(), (uint64_t)totalDumped, blockCnt.as<uint32_t>() );
	hipLaunchKernelGGL( kScanCounts, dim3( 1 ), dim3( 1024 ), 0, st, blockCnt.as<uint32_t>(), (uint64_t)totalDumped, scalarOut );
	uint32_t nVoxels = 0;
	MVRT_HIP( hipMemcpyAsync( &nVoxels, scalarOut, 4, hipMemcpyDeviceToHost, st ) );
	MVRT_HIP( hipStreamSynchronize( st ) );
	Buf morton, attrs;
	if( morton.alloc( (uint64_t)nVoxels * 8 ) || attrs.alloc( (uint64_t)nVoxels * 8 ) ) return 1;
	hipLaunchKernelGGL( kUniqueEmit, dim3( gridFor( totalDumped ) ), dim3( BB ), 0, st, keysB.as<uint64_t>(), valsB.as<uint64_t>(), (uint64_t)totalDumped,
						blockCnt.as<uint32_t>(), morton.as<uint64_t>(), attrs.as<uint2>(), hasEmission );
	MVRT_HIP( hipStreamSynchronize( st ) );
	keysB.release();
	valsB.release();

	// ---- upper bound on nodes: distinct parents per level (octreeTaskInit's taskCounters, voxKernel.cu:257-265) ----
	// counted on the host from the per-level group counts as the levels are built; the node buffer grows by level.
	Buf tasksA, tasksB;
	if( tasksA.alloc( (uint64_t)nVoxels * sizeof( Task ) ) || tasksB.alloc( (uint64_t)nVoxels * sizeof( Task ) ) ) return 1;
	hipLaunchKernelGGL( kInitTasks, dim3( gridFor( nVoxels ) ), dim3( BB ), 0, st, morton.as<uint64_t>(), nVoxels, tasksA.as<Task>() );

	std::vector<Node64*> levelNodes; // per-level node arrays, concatenated at the end
	std::vector<uint8_t*> levelMasks;
	std::vector<uint32_t> levelCount;
	bool tree = false; // no DAG + no embedded masks: compact { mask, first child } nodes + two-level bricks instead of 64-byte lines per node
	Buf treeFirst, bricks;
	uint32_t treeLevelBase[24] = { 0 }, treeLevelCount[24] = { 0 }, treeBrickBase[24] = { 0 }, nBricks = 0;
	auto freeLevels = [&]() {
		for( Node64* p : levelNodes ) (void)hipFree( p );
		for( uint8_t* p : levelMasks ) (void)hipFree( p );
	};

	uint32_t nInput = nVoxels;
	uint32_t nodeBase = 0;
	Task* cur = tasksA.as<Task>();
	Task* nxt = tasksB.as<Task>();
	for( int level = 0; level < levels; level++ )
	{
		// sibling groups
		hipLaunchKernelGGL( kGroupCount, dim3( gridFor( nInput ) ), dim3( BB ), 0, st, cur, (uint64_t)nInput, blockCnt.as<uint32_t>() );
		hipLaunchKernelGGL( kScanCounts, dim3( 1 ), dim3( 1024 ), 0, st, blockCnt.as<uint32_t>(), (uint64_t)nInput, scalarOut );
		uint32_t nGroups = 0;
		MVRT_HIP( hipMemcpyAsync( &nGroups, scalarOut, 4, hipMemcpyDeviceToHost, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );

		if( !dag ) // every group is a node: write it straight into the (pre-sized) final arrays
		{
			if( level == 0 )
			{
				// distinct parents per level in one pass -> exact node count -> one allocation, no concatenation copy
				Buf dCounts;
				if( dCounts.alloc( 32 * 8 ) ) return 1;
				MVRT_HIP( hipMemsetAsync( dCounts.p, 0, 32 * 8, st ) );
				hipLaunchKernelGGL( kLevelCounts, dim3( gridFor( nVoxels ) ), dim3( BB ), 0, st, morton.as<uint64_t>(), (uint64_t)nVoxels, levels, dCounts.as<unsigned long long>() );
				unsigned long long hc[32];
				MVRT_HIP( hipMemcpyAsync( hc, dCounts.p, sizeof( hc ), hipMemcpyDeviceToHost, st ) );
				MVRT_HIP( hipStreamSynchronize( st ) );
				unsigned long long total = 0;
				for( int l = 0; l < levels; l++ ) total += hc[l];
				if( total >= 0xFFFFFFFEull )
				{
					mvrtSetError( "mvrt_svo_build: %llu nodes exceed 32-bit node indices", total );
					return 1;
				}
				tree = total >= 0xFFFFFFull || ( flags & 2 );
				uint8_t* mm = nullptr;
				MVRT_HIP( hipMalloc( (void**)&mm, total ) );
				levelMasks.push_back( mm );
				levelCount.push_back( (uint32_t)total );
				if( tree )
				{
					// node ranges per builder level (0 = parents of voxels ... levels - 1 = root) and brick ranges of the odd levels
					unsigned long long nb = 0, nBr = 0;
					for( int l = 0; l < levels; l++ )
					{
						treeLevelBase[l] = (uint32_t)nb;
						treeLevelCount[l] = (uint32_t)hc[l];
						treeBrickBase[l] = (uint32_t)nBr;
						nb += hc[l];
						if( l & 1 ) nBr += hc[l];
					}
					nBricks = (uint32_t)nBr;
					if( treeFirst.alloc( total * 4 ) || bricks.alloc( ( nBr ? nBr : 1 ) * sizeof( uint4 ) ) ) return 1;
					levelNodes.push_back( nullptr );
				}
				else
				{
					Node64* nn = nullptr;
					MVRT_HIP( hipMalloc( (void**)&nn, total * sizeof( Node64 ) ) );
					levelNodes.push_back( nn );
				}
			}
			if( tree )
			{
				hipLaunchKernelGGL( kMakeNodesTree, dim3( gridFor( nInput ) ), dim3( BB ), 0, st, cur, (uint64_t)nInput, blockCnt.as<uint32_t>(), nodeBase, level == 0 ? 1 : 0, levelMasks[0],
									treeFirst.as<uint32_t>(), nxt );
				if( level & 1 )
					hipLaunchKernelGGL( kMakeBricks, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, levelMasks[0], treeFirst.as<uint32_t>(), treeLevelBase[level], nGroups, level - 1 == 0 ? 1 : 0,
										level >= 2 ? treeLevelBase[level - 2] : 0u, level >= 2 ? treeBrickBase[level - 2] : 0u, bricks.as<uint4>() + treeBrickBase[level] );
			}
			else
				hipLaunchKernelGGL( kMakeNodesDirect, dim3( gridFor( nInput ) ), dim3( BB ), 0, st, cur, (uint64_t)nInput, blockCnt.as<uint32_t>(), nodeBase, levelNodes[0], levelMasks[0], nxt );
			MVRT_HIP( hipStreamSynchronize( st ) );
			nodeBase += nGroups;
			nInput = nGroups;
			Task* t = cur;
			cur = nxt;
			nxt = t;
			continue;
		}
		Buf cands, hashes, groupIds, parents, hashesS, groupS, headPos, headScan, repOf, nodeOfGroup, blockCnt2;
		if( cands.alloc( (uint64_t)nGroups * sizeof( Cand ) ) || hashes.alloc( (uint64_t)nGroups * 8 ) || groupIds.alloc( (uint64_t)nGroups * 4 ) ||
			parents.alloc( (uint64_t)nGroups * 8 ) || hashesS.alloc( (uint64_t)nGroups * 8 ) || groupS.alloc( (uint64_t)nGroups * 4 ) ||
			headPos.alloc( (uint64_t)nGroups * 4 ) || headScan.alloc( (uint64_t)nGroups * 4 ) || repOf.alloc( (uint64_t)nGroups * 4 ) ||
			nodeOfGroup.alloc( (uint64_t)nGroups * 4 ) || blockCnt2.alloc( ( (uint64_t)nGroups / BB + 2 ) * 4 ) )
		{
			freeLevels();
			return 1;
		}
		hipLaunchKernelGGL( kMakeCandidates, dim3( gridFor( nInput ) ), dim3( BB ), 0, st, cur, (uint64_t)nInput, blockCnt.as<uint32_t>(), cands.as<Cand>(),
							hashes.as<uint64_t>(), groupIds.as<uint32_t>(), parents.as<uint64_t>() );

		// DAG de-duplication by sorted content hash
		{
			size_t tmpBytes = 0, tmpBytes2 = 0;
			MVRT_HIP( hipcub::DeviceRadixSort::SortPairs( nullptr, tmpBytes, hashes.as<uint64_t>(), hashesS.as<uint64_t>(), groupIds.as<uint32_t>(), groupS.as<uint32_t>(),
														  nGroups, 0, 64, st ) );
			MVRT_HIP( hipcub::DeviceScan::InclusiveScan( nullptr, tmpBytes2, headPos.as<uint32_t>(), headScan.as<uint32_t>(), MaxOp(), nGroups, st ) );
			Buf tmp;
			if( tmp.alloc( tmpBytes > tmpBytes2 ? tmpBytes : tmpBytes2 ) )
			{
				freeLevels();
				return 1;
			}
			MVRT_HIP( hipcub::DeviceRadixSort::SortPairs( tmp.p, tmpBytes, hashes.as<uint64_t>(), hashesS.as<uint64_t>(), groupIds.as<uint32_t>(), groupS.as<uint32_t>(),
														  nGroups, 0, 64, st ) );
			hipLaunchKernelGGL( kMarkRuns, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, hashesS.as<uint64_t>(), groupS.as<uint32_t>(), cands.as<Cand>(), nGroups,
								headPos.as<uint32_t>() );
			MVRT_HIP( hipcub::DeviceScan::InclusiveScan( tmp.p, tmpBytes2, headPos.as<uint32_t>(), headScan.as<uint32_t>(), MaxOp(), nGroups, st ) );
			hipLaunchKernelGGL( kScatterReps, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, groupS.as<uint32_t>(), headScan.as<uint32_t>(), nGroups, repOf.as<uint32_t>() );
			MVRT_HIP( hipStreamSynchronize( st ) ); // tmp is released at scope end
		}
		// number the first occurrences in group order and write their nodes
		hipLaunchKernelGGL( kFirstCount, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, repOf.as<uint32_t>(), (uint64_t)nGroups, blockCnt2.as<uint32_t>() );
		hipLaunchKernelGGL( kScanCounts, dim3( 1 ), dim3( 1024 ), 0, st, blockCnt2.as<uint32_t>(), (uint64_t)nGroups, scalarOut );
		uint32_t nUnique = 0;
		MVRT_HIP( hipMemcpyAsync( &nUnique, scalarOut, 4, hipMemcpyDeviceToHost, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );
		Node64* lvNodes = nullptr;
		uint8_t* lvMasks = nullptr;
		MVRT_HIP( hipMalloc( (void**)&lvNodes, (uint64_t)nUnique * sizeof( Node64 ) ) );
		MVRT_HIP( hipMalloc( (void**)&lvMasks, nUnique ) );
		levelNodes.push_back( lvNodes );
		levelMasks.push_back( lvMasks );
		levelCount.push_back( nUnique );
		// kEmitNodes writes to nodes[nodeBase + rank]: pass pointers rebased so that index nodeBase lands on lvNodes[0]
		hipLaunchKernelGGL( kEmitNodes, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, repOf.as<uint32_t>(), cands.as<Cand>(), (uint64_t)nGroups, blockCnt2.as<uint32_t>(),
							nodeBase, lvNodes - nodeBase, lvMasks - nodeBase, nodeOfGroup.as<uint32_t>() );
		hipLaunchKernelGGL( kNextTasks, dim3( gridFor( nGroups ) ), dim3( BB ), 0, st, repOf.as<uint32_t>(), nodeOfGroup.as<uint32_t>(), cands.as<Cand>(),
							parents.as<uint64_t>(), nGroups, nxt );
		MVRT_HIP( hipStreamSynchronize( st ) );
		nodeBase += nUnique;
		nInput = nGroups;
		Task* t = cur;
		cur = nxt;
		nxt = t;
	}
	if( nInput != 1 )
	{
		freeLevels();
		mvrtSetError( "mvrt_svo_build: internal error, %u roots after %d levels", nInput, levels );
		return 1;
	}
	const uint32_t nNodes = nodeBase;
	Buf nodes, masks;
	if( !dag )
	{
		nodes.p = levelNodes[0]; // written in place, already final
		masks.p = levelMasks[0];
		levelNodes.clear();
		levelMasks.clear();
	}
	else if( nodes.alloc( (uint64_t)nNodes * sizeof( Node64 ) ) || masks.alloc( nNodes ) )
	{
		freeLevels();
		return 1;
	}
	if( dag )
	{
		uint64_t off = 0;
		for( size_t l = 0; l < levelNodes.size(); l++ )
		{
			MVRT_HIP( hipMemcpyAsync( nodes.as<Node64>() + off, levelNodes[l], (uint64_t)levelCount[l] * sizeof( Node64 ), hipMemcpyDeviceToDevice, st ) );
			MVRT_HIP( hipMemcpyAsync( masks.as<uint8_t>() + off, levelMasks[l], levelCount[l], hipMemcpyDeviceToDevice, st ) );
			off += levelCount[l];
		}
		MVRT_HIP( hipStreamSynchronize( st ) );
		freeLevels();
	}
	if( tree ) // nothing left to convert: the traversal reads the bricks, download rebuilds the reference's nodes from { mask, first child }
	{
		uint32_t he = 0;
		MVRT_HIP( hipMemcpyAsync( &he, hasEmission, 4, hipMemcpyDeviceToHost, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );
		MVRT_HIP( hipGetLastError() );
		out->nodes = (Node64*)bricks.detach();
		out->masks = (uint8_t*)masks.detach();
		out->psumCold = nullptr;
		out->treeFirst = (uint32_t*)treeFirst.detach();
		out->tree = 1;
		out->nBricks = nBricks;
		for( int l = 0; l < 24; l++ )
		{
			out->treeLevelBase[l] = treeLevelBase[l];
			out->treeLevelCount[l] = treeLevelCount[l];
		}
		// where the traversal starts: the root is the single node of builder level levels - 1.  Odd level: it is a brick root -> its brick.
		// Even level: it is an "in-brick child" whose children (level levels - 2) are brick roots -> the first of them (or, for a one-level
		// octree, voxel 0)
		const int rl = levels - 1;
		out->treeRoot = ( rl & 1 ) ? treeBrickBase[rl] : ( rl == 0 ? 0u : treeBrickBase[rl - 1] );
		out->attrs = (uint2*)attrs.detach();
		out->morton = (uint64_t*)morton.detach();
		out->nNodes = nNodes;
		out->nVoxels = nVoxels;
		out->hasEmission = he;
		out->embedded = 0;
		out->totalDumped = totalDumped;
		return 0;
	}
	const bool embed = nNodes < 0xFFFFFFu && !( flags & 2 );
	if( embed )
	{
		hipLaunchKernelGGL( kEmbedMasks, dim3( divUp( (uint64_t)nNodes * 8, BB ) ), dim3( BB ), 0, st, nodes.as<Node64>(), masks.as<uint8_t>(), nNodes );
	}
	Buf psumCold;
	if( !embed ) // non-embedded flavour: nVoxelsPSum -> cold array, child masks -> the hot line (traverse_stream.h)
	{
		if( psumCold.alloc( (uint64_t)nNodes * 32 ) ) return 1;
		if( launchSplitPsum( nodes.as<Node64>(), masks.as<uint8_t>(), psumCold.as<uint32_t>(), nNodes, st ) ) return 1;
	}
	uint32_t he = 0;
	MVRT_HIP( hipMemcpyAsync( &he, hasEmission, 4, hipMemcpyDeviceToHost, st ) );
	MVRT_HIP( hipStreamSynchronize( st ) );
	MVRT_HIP( hipGetLastError() );

	out->nodes = (Node64*)nodes.detach();
	out->masks = (uint8_t*)masks.detach();
	out->psumCold = (uint32_t*)psumCold.detach();
	out->attrs = (uint2*)attrs.detach();
	out->morton = (uint64_t*)morton.detach();
	out->nNodes = nNodes;
	out->nVoxels = nVoxels;
	out->hasEmission = he;
	out->embedded = embed ? 1 : 0;
	out->totalDumped = totalDumped;
	return 0;
}
