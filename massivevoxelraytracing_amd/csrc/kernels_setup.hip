// kernels_setup.hip -- kernels around the hot path: node layout conversion, HDRI summed-area tables
// (voxKernel.cu:485-608 semantics), resolve (voxKernel.cu:779-795), multi-GPU tile assembly.
#include "launch.h"

// ---- reference 68-byte AoS nodes <-> one 64-byte line per node -----------------------------------------
__global__ void __launch_bounds__( 256 ) kConvertNodes( const uint8_t* __restrict__ nodes68, uint32_t nNodes, Node64* __restrict__ out, uint8_t* __restrict__ masks,
														 uint32_t* __restrict__ psumCold, int nonEmbedded )
{
	// one thread per (node, dword): 16 payload dwords per node
	uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t node = gid / 16;
	uint32_t w = gid % 16;
	if( node >= nNodes ) return;
	const uint32_t* src = (const uint32_t*)( nodes68 + node * 68 ); // 68 is a multiple of 4: aligned dwords
	const uint32_t v = src[1 + w];
	if( nonEmbedded && w >= 8 ) psumCold[node * 8 + ( w - 8 )] = v; // the hot line keeps child masks instead (kFillChildMasks)
	else ( (uint32_t*)( out + node ) )[w] = v;
	if( w == 0 ) masks[node] = (uint8_t)( src[0] & 0xFF );
}
// non-embedded flavour: the 8 child masks of every node, one byte each, into psum[0..1] of its line
__global__ void __launch_bounds__( 256 ) kFillChildMasks( Node64* __restrict__ nodes, const uint8_t* __restrict__ masks, uint64_t nNodes )
{
	for( uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nNodes; i += (uint64_t)gridDim.x * 256 )
	{
		uint32_t lo = 0, hi = 0;
		for( int k = 0; k < 8; k++ )
		{
			const uint32_t c = nodes[i].children[k];
			const uint32_t m = c == MVRT_LEAF ? 0u : (uint32_t)masks[c];
			if( k < 4 ) lo |= m << ( 8 * k );
			else hi |= m << ( 8 * ( k - 4 ) );
		}
		nodes[i].psum[0] = lo;
		nodes[i].psum[1] = hi;
	}
}
__global__ void __launch_bounds__( 256 ) kMovePsumCold( const Node64* __restrict__ nodes, uint32_t* __restrict__ psumCold, uint64_t nNodes )
{
	for( uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < nNodes * 8; g += (uint64_t)gridDim.x * 256 ) psumCold[g] = nodes[g / 8].psum[g % 8];
}
static uint32_t cappedGrid( uint64_t n )
{
	uint64_t b = ( n + 255 ) / 256;
	if( b < 1 ) b = 1;
	return (uint32_t)( b > 65536 ? 65536 : b );
}
int launchConvertNodes( const uint8_t* nodes68, uint32_t nNodes, Node64* out, uint8_t* masks, uint32_t* psumCold, int nonEmbedded, hipStream_t stream )
{
	if( nNodes == 0 ) return 0;
	hipLaunchKernelGGL( kConvertNodes, dim3( divUp( (uint64_t)nNodes * 16, 256 ) ), dim3( 256 ), 0, stream, nodes68, nNodes, out, masks, psumCold, nonEmbedded );
	if( nonEmbedded ) hipLaunchKernelGGL( kFillChildMasks, dim3( cappedGrid( nNodes ) ), dim3( 256 ), 0, stream, out, masks, (uint64_t)nNodes );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
// embedded flavour, uploaded octrees: is nVoxelsPSum of every PARENT OF VOXELS the exclusive popcount of its mask (what bottomUpOctreeBuild writes,
// voxKernel.cu:296-329)?  voxelIndexFromPath replaces the last level's nVoxelsPSum fetch by that popcount only then; an octree uploaded with other
// values there (e.g. buildOctreeNaive's zeros, IntersectorOctree.hpp:195) gets the stored values, as the reference's traversal would use them.
__global__ void __launch_bounds__( 256 ) kCheckLeafPsum( const Node64* __restrict__ nodes, const uint8_t* __restrict__ masks, uint32_t nNodes, uint32_t* __restrict__ bad )
{
	for( uint32_t n = blockIdx.x * 256 + threadIdx.x; n < nNodes; n += gridDim.x * 256 )
	{
		const uint32_t mask = masks[n];
		uint32_t run = 0;
		bool wrong = false;
		for( uint32_t c = 0; c < 8; c++ )
		{
			if( !( ( mask >> c ) & 1u ) ) continue;
			if( nodes[n].children[c] == MVRT_LEAF && nodes[n].psum[c] != run ) wrong = true;
			run++;
		}
		if( wrong ) atomicOr( bad, 1u );
	}
}
int launchCheckLeafPsum( const Node64* nodes, const uint8_t* masks, uint32_t nNodes, uint32_t* badDev, hipStream_t stream )
{
	MVRT_HIP( hipMemsetAsync( badDev, 0, 4, stream ) );
	hipLaunchKernelGGL( kCheckLeafPsum, dim3( cappedGrid( nNodes ) ), dim3( 256 ), 0, stream, nodes, masks, nNodes, badDev );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
int launchSplitPsum( Node64* nodes, const uint8_t* masks, uint32_t* psumCold, uint64_t nNodes, hipStream_t stream )
{
	if( nNodes == 0 ) return 0;
	hipLaunchKernelGGL( kMovePsumCold, dim3( cappedGrid( nNodes * 8 ) ), dim3( 256 ), 0, stream, nodes, psumCold, nNodes );
	hipLaunchKernelGGL( kFillChildMasks, dim3( cappedGrid( nNodes ) ), dim3( 256 ), 0, stream, nodes, masks, nNodes );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
// ---- tree flavour -> reference layout (mvrt_svo_download).  Builder level 0 = parents of voxels, levels - 1 = root; nodes of a level are in
// morton order, their children consecutive in the level below, the voxels consecutive in morton order: a node covers a contiguous voxel range.
struct TreeLevels
{
	uint32_t base[24], count[24];
	int levels;
};
MVRT_DI uint32_t treeFirstVoxel( const uint32_t* __restrict__ first, uint32_t idx, int l ) // first voxel below node idx of builder level l
{
	for( ; l > 0; l-- ) idx = first[idx];
	return first[idx];
}
__global__ void __launch_bounds__( 256 ) kTreeTo68( const uint8_t* __restrict__ masks, const uint32_t* __restrict__ first, TreeLevels L, uint32_t nNodes, uint32_t nVoxels,
													 uint8_t* __restrict__ nodes68 )
{
	for( uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < nNodes; idx += (uint64_t)gridDim.x * 256 )
	{
		int l = 0;
		while( l + 1 < L.levels && idx >= L.base[l + 1] ) l++;
		const uint32_t mask = masks[idx];
		uint32_t* dst = (uint32_t*)( nodes68 + idx * 68 );
		dst[0] = mask;
		uint32_t k = first[idx], run = 0; // child nodes (l > 0) or voxels (l == 0), consecutive from k
		const uint32_t myVox = l > 0 ? treeFirstVoxel( first, k, l - 1 ) : k;
		for( int c = 0; c < 8; c++ )
		{
			uint32_t child = MVRT_LEAF, cnt = 0;
			if( mask >> c & 1u )
			{
				if( l == 0 ) cnt = 1;
				else
				{
					child = k;
					const uint32_t endLevel = L.base[l - 1] + L.count[l - 1];
					const uint32_t nextVox = k + 1 < endLevel ? treeFirstVoxel( first, k + 1, l - 1 ) : nVoxels;
					cnt = nextVox - ( myVox + run );
				}
				k++;
			}
			dst[1 + c] = child;
			dst[9 + c] = run; // nVoxelsPSum: voxels below the children before c (exclusive prefix, also for absent children)
			run += cnt;
		}
	}
}
int launchTreeTo68( const uint8_t* masks, const uint32_t* first, const uint32_t* levelBase, const uint32_t* levelCount, int levels, uint32_t nNodes, uint32_t nVoxels, uint8_t* nodes68,
					hipStream_t stream )
{
	TreeLevels L;
	for( int l = 0; l < 24; l++ )
	{
		L.base[l] = levelBase[l];
		L.count[l] = levelCount[l];
	}
	L.levels = levels;
	hipLaunchKernelGGL( kTreeTo68, dim3( cappedGrid( nNodes ) ), dim3( 256 ), 0, stream, masks, first, L, nNodes, nVoxels, nodes68 );
	MVRT_HIP( hipGetLastError() );
	return 0;
}

// topTable (SvoDev): one thread per path prefix of `k` levels walks it from the root exactly as voxelIndexFromPath would
__global__ void __launch_bounds__( 256 ) kBuildTopTable( const Node64* __restrict__ nodes, uint32_t rootIndex, uint32_t k, uint2* __restrict__ table )
{
	const uint32_t nPrefix = 1u << ( 3u * k );
	for( uint32_t p = blockIdx.x * 256 + threadIdx.x; p < nPrefix; p += gridDim.x * 256 )
	{
		uint32_t n = rootIndex, v = 0;
		bool alive = true;
		for( uint32_t l = 0; l < k && alive; l++ )
		{
			const uint32_t c = ( p >> ( 3u * ( k - 1u - l ) ) ) & 7u;
			const uint32_t child = nodes[n].children[c];
			v += nodes[n].psum[c];
			if( child == MVRT_LEAF ) alive = l + 1 == k; // a voxel (only legal at the last level of the octree) or an absent child
			n = child & 0xFFFFFFu;
		}
		table[p] = alive ? make_uint2( n, v ) : make_uint2( 0u, 0u );
	}
}
__global__ void __launch_bounds__( 256 ) kCopyKids( const Node64* __restrict__ nodes, uint64_t nNodes, uint32_t* __restrict__ kids )
{
	for( uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x; g < nNodes * 8; g += (uint64_t)gridDim.x * 256 ) kids[g] = nodes[g / 8].children[g % 8];
}
int launchCopyKids( const Node64* nodes, uint64_t nNodes, uint32_t* kids, hipStream_t stream )
{
	hipLaunchKernelGGL( kCopyKids, dim3( cappedGrid( nNodes * 8 ) ), dim3( 256 ), 0, stream, nodes, nNodes, kids );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
// prefix tables of the start below the root (traverse_stream.h): one thread per (level l, path prefix p of l levels) walks p from the root through
// the children array and stores the reference it arrives at (index | mask << 24; 0 when the prefix leaves the octree: never looked up)
__global__ void __launch_bounds__( 256 ) kBuildPrefixRefs( const uint32_t* __restrict__ kids, uint32_t rootRef, uint32_t tabLevels, uint32_t* __restrict__ table )
{
	const uint32_t total = 0x49249249u & ( ( 1u << ( 3u * ( tabLevels + 1u ) ) ) - 1u ); // sum of 8^l, l = 0..tabLevels
	for( uint32_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256 )
	{
		uint32_t l = 0;
		while( ( 0x49249249u & ( ( 1u << ( 3u * ( l + 1u ) ) ) - 1u ) ) <= e ) l++; // level of entry e
		const uint32_t p = e - ( 0x49249249u & ( ( 1u << ( 3u * l ) ) - 1u ) );
		uint32_t ref = rootRef;
		bool alive = true;
		for( uint32_t k = 0; k < l && alive; k++ )
		{
			const uint32_t c = ( p >> ( 3u * ( l - 1u - k ) ) ) & 7u;
			if( !( ( ref >> ( 24u + c ) ) & 1u ) ) alive = false;
			else ref = kids[( ref & 0xFFFFFFu ) * 8u + c];
			if( ref == MVRT_LEAF ) alive = false; // a voxel: prefixes never reach the voxel level (tabLevels <= levels - 1), but uploaded octrees may be shallower in places
		}
		table[e] = alive ? ref : 0u;
	}
}
int launchBuildPrefixRefs( const uint32_t* kids, uint32_t rootRef, uint32_t tabLevels, uint32_t* table, hipStream_t stream )
{
	const uint32_t total = 0x49249249u & ( ( 1u << ( 3u * ( tabLevels + 1u ) ) ) - 1u );
	hipLaunchKernelGGL( kBuildPrefixRefs, dim3( cappedGrid( total ) ), dim3( 256 ), 0, stream, kids, rootRef, tabLevels, table );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
// cell index (mvrt_common.h SvoDev::cellBlocks / cellEntries) from the sorted voxel codes of a build.  shift = 3 + cellBits: a voxel code without its last
// `shift` bits is its block.  Pass 1 numbers the occupied blocks (the first voxel of a block takes the next number: which block gets which number depends on
// the order the waves arrive in, what is stored under it does not); pass 2 writes the entry of every occupied cell (the thread of a cell's first voxel owns it).
__global__ void __launch_bounds__( 256 ) kNumberCellBlocks( const uint64_t* __restrict__ morton, uint64_t n, uint32_t shift, uint32_t* __restrict__ blocks, uint32_t* __restrict__ counter )
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if( i >= n ) return;
	const uint64_t b = morton[i] >> shift;
	if( i != 0 && ( morton[i - 1] >> shift ) == b ) return;
	blocks[b] = atomicAdd( counter, 1u );
}
__global__ void __launch_bounds__( 256 ) kFillCellIndex( const uint64_t* __restrict__ morton, uint64_t n, uint32_t cellBits, const uint32_t* __restrict__ blocks, uint2* __restrict__ entries )
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if( i >= n ) return;
	const uint64_t cell = morton[i] >> 3;
	if( i != 0 && ( morton[i - 1] >> 3 ) == cell ) return;
	uint32_t mask = 0;
	for( uint64_t j = i; j < n && j < i + 8; j++ )
	{
		const uint64_t c = morton[j];
		if( ( c >> 3 ) != cell ) break;
		mask |= 1u << ( (uint32_t)c & 7u );
	}
	// rank of the cell's first voxel = what nVoxelsPSum adds up to along its path; the mask ranks its voxels
	entries[( (uint64_t)blocks[cell >> cellBits] << cellBits ) | ( (uint32_t)cell & ( ( 1u << cellBits ) - 1u ) )] = make_uint2( (uint32_t)i, mask );
}
int launchNumberCellBlocks( const uint64_t* morton, uint64_t n, uint32_t cellBits, uint32_t* blocks, uint32_t* counterDev, hipStream_t stream )
{
	if( n == 0 ) return 0;
	hipLaunchKernelGGL( kNumberCellBlocks, dim3( divUp( n, 256 ) ), dim3( 256 ), 0, stream, morton, n, 3u + cellBits, blocks, counterDev );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
int launchFillCellIndex( const uint64_t* morton, uint64_t n, uint32_t cellBits, const uint32_t* blocks, uint2* entries, hipStream_t stream )
{
	if( n == 0 ) return 0;
	hipLaunchKernelGGL( kFillCellIndex, dim3( divUp( n, 256 ) ), dim3( 256 ), 0, stream, morton, n, cellBits, blocks, entries );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
int launchBuildTopTable( const Node64* nodes, uint32_t rootIndex, uint32_t k, uint2* table, hipStream_t stream )
{
	hipLaunchKernelGGL( kBuildTopTable, dim3( cappedGrid( 1ull << ( 3 * k ) ) ), dim3( 256 ), 0, stream, nodes, rootIndex, k, table );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
__global__ void __launch_bounds__( 256 ) kNodesTo68( const Node64* __restrict__ nodes, const uint8_t* __restrict__ masks, const uint32_t* __restrict__ psumCold, uint32_t nNodes,
													  uint8_t* __restrict__ nodes68, int nonEmbedded )
{
	uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t node = gid / 17;
	uint32_t w = gid % 17;
	if( node >= nNodes ) return;
	uint32_t* dst = (uint32_t*)( nodes68 + node * 68 );
	uint32_t v = w == 0 ? (uint32_t)masks[node] : ( (const uint32_t*)( nodes + node ) )[w - 1];
	if( nonEmbedded && w >= 9 ) v = psumCold[node * 8 + ( w - 9 )];
	dst[w] = v;
}
int launchNodesTo68( const Node64* nodes, const uint8_t* masks, const uint32_t* psumCold, uint32_t nNodes, uint8_t* nodes68, int nonEmbedded, hipStream_t stream )
{
	if( nNodes == 0 ) return 0;
	hipLaunchKernelGGL( kNodesTo68, dim3( divUp( (uint64_t)nNodes * 17, 256 ) ), dim3( 256 ), 0, stream, nodes, masks, psumCold, nNodes, nodes68, nonEmbedded );
	MVRT_HIP( hipGetLastError() );
	return 0;
}

// ---- HDRI importance tables ------------------------------------------------------------------------------
// HDRIstoreImportance, voxKernel.cu:485-524
__global__ void kHdriImportance( const float4* __restrict__ pixels, int w, int h, double* __restrict__ sat, int cosWeighted, f3 axis )
{
	uint32_t pixelX = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t pixelY = blockIdx.y * blockDim.y + threadIdx.y;
	if( (uint32_t)w <= pixelX || (uint32_t)h <= pixelY ) return;
	uint32_t pixelIdx = pixelY * w + pixelX;
	float dTheta = MVRT_PI / (float)h;
	float dPhi = 2.0f * MVRT_PI / (float)w;
	float theta = pixelY * dTheta;
	float dH = 2.0f * mvrt_sin( dTheta * 0.5f ) * mvrt_sin( dTheta * 0.5f + theta );
	float sr = dH * dPhi;
	float4 color = pixels[pixelIdx];
	float wgt = 1.0f;
	if( cosWeighted )
	{
		float sY = mixf( mvrt_cos( theta ), mvrt_cos( theta + dTheta ), 0.5f );
		float phi = dPhi * ( (float)pixelX + 0.5f ) + MVRT_PI;
		float sX = mvrt_cos( phi );
		float sZ = mvrt_sin( phi );
		float sinTheta = sqrtf( smax( 1.0f - sY * sY, 0.0f ) );
		f3 dirCenter = mk3( sX * sinTheta, sY, sZ * sinTheta );
		wgt = smax( dot3( axis, dirCenter ), 0.0f );
	}
	float lum = 0.2126f * color.x + 0.7152f * color.y + 0.0722f * color.z; // luminance, renderCommon.hpp:168-172
	sat[pixelIdx] = lum * sr * wgt;
}

// Inclusive prefix sums of rows (horizontal != 0) or columns in f64.  The ADDITION ORDER is part of the
// result (f64 rounding feeds the u32 tables), so it is fixed to the reference's: 512-wide chunks, each
// scanned Hillis-Steele in LDS, carried by a running prefix (voxKernel.cu:526-598).
#define SAT_BLOCK 512
__global__ void __launch_bounds__( SAT_BLOCK ) kSatScan( int w, int h, double* __restrict__ sat, int horizontal )
{
	__shared__ double s[SAT_BLOCK];
	const int line = blockIdx.x;
	const int len = horizontal ? w : h;
	double prefix = 0.0;
	for( int i = 0; i < len; i += SAT_BLOCK )
	{
		const int k = i + threadIdx.x;
		const uint64_t idx = horizontal ? (uint64_t)line * w + k : (uint64_t)k * w + line;
		s[threadIdx.x] = k < len ? sat[idx] : 0.0;
		__syncthreads();
		for( uint32_t off = 1; off < SAT_BLOCK; off <<= 1 )
		{
			double x = s[threadIdx.x];
			if( off <= threadIdx.x ) x += s[threadIdx.x - off];
			__syncthreads();
			s[threadIdx.x] = x;
			__syncthreads();
		}
		double sum = s[SAT_BLOCK - 1];
		__syncthreads();
		s[threadIdx.x] += prefix;
		__syncthreads();
		prefix += sum;
		if( k < len ) sat[idx] = s[threadIdx.x];
		__syncthreads();
	}
}
// buildSAT2u32, voxKernel.cu:600-608
__global__ void kSatToU32( uint32_t* __restrict__ satU32, const double* __restrict__ satF64, int n )
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	double sum = satF64[n - 1];
	if( i < n ) satU32[i] = (uint32_t)( satF64[i] / ( sum ) * (double)0xFFFFFFFFu );
}
int launchHdriSat( const float4* pixels, int w, int h, double* satF64, uint32_t* satOut, int cosWeighted, f3 axis, hipStream_t stream )
{
	hipLaunchKernelGGL( kHdriImportance, dim3( divUp( w, 8 ), divUp( h, 8 ) ), dim3( 8, 8 ), 0, stream, pixels, w, h, satF64, cosWeighted, axis );
	hipLaunchKernelGGL( kSatScan, dim3( h ), dim3( SAT_BLOCK ), 0, stream, w, h, satF64, 1 );
	hipLaunchKernelGGL( kSatScan, dim3( w ), dim3( SAT_BLOCK ), 0, stream, w, h, satF64, 0 );
	hipLaunchKernelGGL( kSatToU32, dim3( divUp( (uint64_t)w * h, 64 ) ), dim3( 64 ), 0, stream, satOut, satF64, w * h );
	MVRT_HIP( hipGetLastError() );
	return 0;
}

// ---- renderResolve, voxKernel.cu:779-795 ------------------------------------------------------------------
__global__ void __launch_bounds__( 256 ) kResolve( const float4* __restrict__ fb, uint64_t n, uchar4* __restrict__ out )
{
	for( uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256 )
	{
		float4 v = fb[i];
		int r = (int)( 255 * mvrt_pow( v.x / v.w, 1.0f / 2.2f ) + 0.5f );
		int g = (int)( 255 * mvrt_pow( v.y / v.w, 1.0f / 2.2f ) + 0.5f );
		int b = (int)( 255 * mvrt_pow( v.z / v.w, 1.0f / 2.2f ) + 0.5f );
		out[i] = make_uchar4( (uint8_t)( r < 255 ? r : 255 ), (uint8_t)( g < 255 ? g : 255 ), (uint8_t)( b < 255 ? b : 255 ), 255 );
	}
}
int launchResolve( const float4* fb, uint64_t n, uchar4* out, hipStream_t stream )
{
	if( n == 0 ) return 0;
	uint32_t grid = divUp( n, 256 );
	if( grid > 2048 ) grid = 2048;
	hipLaunchKernelGGL( kResolve, dim3( grid ), dim3( 256 ), 0, stream, fb, n, out );
	MVRT_HIP( hipGetLastError() );
	return 0;
}

// ---- tile assembly after the RCCL all-gather --------------------------------------------------------------
// gathered: tileCount buffers of rankStridePixels float4, each holding that rank's owned pixels in block
// order (block b of the frame -> rank b % tileCount, local block b / tileCount).
__global__ void __launch_bounds__( 256 ) kAssembleTiles( const float4* __restrict__ gathered, int tileCount, uint64_t rankStride, uint64_t nPixels, float4* __restrict__ frame )
{
	for( uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < nPixels; p += (uint64_t)gridDim.x * 256 )
	{
		uint64_t b = p / MVRT_TILE_PIXELS;
		uint64_t rank = b % tileCount;
		uint64_t local = ( b / tileCount ) * MVRT_TILE_PIXELS + ( p % MVRT_TILE_PIXELS );
		frame[p] = gathered[rank * rankStride + local];
	}
}
int launchAssembleTiles( const float4* gathered, int tileCount, uint64_t rankStridePixels, int W, int H, float4* frame, hipStream_t stream )
{
	uint64_t n = (uint64_t)W * H;
	if( n == 0 ) return 0;
	uint32_t grid = divUp( n, 256 );
	if( grid > 2048 ) grid = 2048;
	hipLaunchKernelGGL( kAssembleTiles, dim3( grid ), dim3( 256 ), 0, stream, gathered, tileCount, rankStridePixels, n, frame );
	MVRT_HIP( hipGetLastError() );
	return 0;
}
