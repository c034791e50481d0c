// kernels_rt.hip -- the hot path: octree traversal, primary cast, wavefront path tracer with stable
// live-path compaction.  Hand-written for gfx950 (wave64, LDS stacks, SoA buffers).
//
// Reference semantics (what each kernel must reproduce): /root/reference/voxKernel.cu:437-483 (render),
// :610-777 (renderPT), voxCommon.hpp:231-423 (traversal), StreamCompaction.hpp:87-184 (stable compaction).
#include <stdlib.h>

#include "launch.h"
#include "traverse_stream.h"

#define WAVE 64
#define CBLOCK 256 // paths per compaction block

MVRT_DI uint32_t laneId() { return __builtin_amdgcn_mbcnt_hi( ~0u, __builtin_amdgcn_mbcnt_lo( ~0u, 0u ) ); }

MVRT_DI unsigned long long waveSum( unsigned long long v )
{
	for( int o = 32; o > 0; o >>= 1 ) v += __shfl_down( v, o, WAVE );
	return v;
}

static int numCUs();

// =====================================================================================================
// Persistent-wave ("stream") variants for embedded-mask octrees: traverse_stream.h
// =====================================================================================================
#define STREAM_WAVES_PER_CU 32

struct BatchIO
{
	const float *rox, *roy, *roz, *rdx, *rdy, *rdz;
	const uint8_t* isShadow;
	float* tOut;
	int32_t* nMajorOut;
	uint64_t* pathOut; // scratch: resolved to vIndex by kResolveVIndex
	uint32_t* descentsOut;
#ifdef MVRT_UTIL_STATS
	unsigned long long utilIters = 0, utilActive = 0, utilTailIters = 0, utilTailActive = 0, utilRefillClocks = 0, utilTotalClocks = 0;
	uint32_t utilMaxRayIters = 0;
#endif
	const uint64_t* originPath; // optional: per ray, the full path of an existing voxel (start-below-the-root hint), ~0 = none
	uint32_t levels;
	MVRT_DI bool load( uint32_t i, f3* ro, f3* rd, uint32_t* hint ) const
	{
		*ro = mk3( rox[i], roy[i], roz[i] );
		*rd = mk3( rdx[i], rdy[i], rdz[i] );
		if( originPath )
		{
			const uint64_t op = originPath[i];
			if( op != ~0ull ) *hint = hintFromVoxelPath( op, levels );
		}
		return isShadow ? isShadow[i] != 0 : false;
	}
	MVRT_DI void store( uint32_t i, const StreamHit& h, bool ) const
	{
		tOut[i] = h.t;
		if( nMajorOut ) nMajorOut[i] = h.nMajor;
		if( pathOut ) pathOut[i] = h.path;
		if( descentsOut ) descentsOut[i] = h.descents;
	}
};
// dense post-pass: vIndex (and optionally the primary-cast colour) from the recorded voxel paths
__global__ void __launch_bounds__( 256 ) kResolveVIndex( SvoDev svo, uint64_t n, const float* __restrict__ t, const uint64_t* __restrict__ path, const uint8_t* __restrict__ isShadow,
														  uint32_t* __restrict__ vIndexOut, uchar4* __restrict__ colorOut )
{
	for( uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256 )
	{
		uint32_t v = 0;
		const bool hit = t[i] != MVRT_MAXF;
		if( hit && !( isShadow && isShadow[i] ) ) v = voxelIndexFromPath( svo, path[i] );
		if( vIndexOut ) vIndexOut[i] = v;
		if( colorOut && hit )
		{
			uint32_t col = svo.attrs[v].x; // IntersectorOctreeGPU::getVoxelColor, voxKernel.cu:466
			colorOut[i] = make_uchar4( col & 0xFF, ( col >> 8 ) & 0xFF, ( col >> 16 ) & 0xFF, ( col >> 24 ) & 0xFF );
		}
	}
}
#ifndef MVRT_TRACE_WAVES
#define MVRT_TRACE_WAVES 7 // waves per SIMD the traversal kernels are register-budgeted for (72 VGPRs)
#endif
// the embedded flavour's 8 KiB LDS ring admits 5 waves per SIMD (5, 6 and 7 perform alike on cache-resident octrees, profiles/r02_traversal_experiments.txt)
#ifndef MVRT_TREE_WAVES
#define MVRT_TREE_WAVES ( MVRT_RING_TREE >= 8 ? 3 : 6 )
#endif
// ... and the tree flavour's 12 KiB per wave (8-slot ring + two mask words per slot) 3
#define MVRT_WAVES_OF( FL ) ( ( FL ) == 0 && MVRT_RING_EMBED >= 8 && MVRT_TRACE_WAVES > 40 / MVRT_RING_EMBED ? 40 / MVRT_RING_EMBED : ( ( FL ) == 2 && MVRT_TRACE_WAVES > MVRT_TREE_WAVES ? MVRT_TREE_WAVES : MVRT_TRACE_WAVES ) )
template <int FL>
__global__ void __launch_bounds__( 64, MVRT_WAVES_OF( FL ) ) kTraceBatchStream( SvoDev svo, uint64_t n, BatchIO io, TraceWorkspace ws, uint32_t chunk )
{
	__shared__ uint4 ring[MVRT_RING_OF( FL ) * 64];
	__shared__ uint32_t ringMask[FL == 0 ? 1 : ( FL == 2 ? 2 : 1 ) * MVRT_RING_OF( FL ) * 64]; // (tree flavour: two mask words per stacked node)
	traceStream<FL>( makeTraceCore( svo ), io, n, ws.cursor, chunk, ring, ws.spill, ws.spillStride, (uint64_t)blockIdx.x * 64 + threadIdx.x, ringMask, ws.spillMask, ws.spillMask2 );
}

struct PrimaryIO
{
	SvoDev svo;
	CameraPinhole cam;
	int W, H, showVertexColor;
	uchar4* rgba;
	float* tOut;
	int32_t* nMajorOut;
	uint64_t* pathOut;
	uint32_t* descentsOut;
#ifdef MVRT_UTIL_STATS
	unsigned long long utilIters = 0, utilActive = 0, utilTailIters = 0, utilTailActive = 0, utilRefillClocks = 0, utilTotalClocks = 0;
	uint32_t utilMaxRayIters = 0;
#endif
	f3 rdKeep; // direction of the ray this lane is tracing (for the normal colour)
	MVRT_DI bool load( uint32_t pixelIdx, f3* ro, f3* rd, uint32_t* )
	{
		int x = (int)( pixelIdx % W );
		int y = (int)( pixelIdx / W );
		cameraShoot( cam, ro, rd, x, y, 0.5f, 0.5f, W, H ); // voxKernel.cu:455
		rdKeep = *rd;
		return false;
	}
	MVRT_DI void store( uint32_t pixelIdx, const StreamHit& r, bool ) const
	{
		uchar4 c = make_uchar4( 0, 0, 0, 255 );
		if( r.t != MVRT_MAXF ) // voxKernel.cu:462-478
		{
			if( !showVertexColor ) // the vertex-colour case is filled in by kResolveVIndex
			{
				f3 hn = getHitN( r.nMajor, rdKeep );
				f3 color = ( hn + mk3( 1.0f, 1.0f, 1.0f ) ) * 0.5f;
				c = make_uchar4( (uint8_t)( 255 * color.x + 0.5f ), (uint8_t)( 255 * color.y + 0.5f ), (uint8_t)( 255 * color.z + 0.5f ), 255 );
			}
		}
		if( rgba ) rgba[pixelIdx] = c;
		if( tOut ) tOut[pixelIdx] = r.t;
		if( nMajorOut ) nMajorOut[pixelIdx] = r.nMajor;
		if( pathOut ) pathOut[pixelIdx] = r.path;
		if( descentsOut ) descentsOut[pixelIdx] = r.descents;
	}
};
template <int FL>
__global__ void __launch_bounds__( 64, MVRT_WAVES_OF( FL ) ) kRenderPrimaryStream( PrimaryIO io, TraceWorkspace ws, uint32_t chunk )
{
	__shared__ uint4 ring[MVRT_RING_OF( FL ) * 64];
	__shared__ uint32_t ringMask[FL == 0 ? 1 : ( FL == 2 ? 2 : 1 ) * MVRT_RING_OF( FL ) * 64]; // (tree flavour: two mask words per stacked node)
	traceStream<FL>( makeTraceCore( io.svo ), io, (uint64_t)io.W * io.H, ws.cursor, chunk, ring, ws.spill, ws.spillStride, (uint64_t)blockIdx.x * 64 + threadIdx.x, ringMask,
						ws.spillMask, ws.spillMask2 );
}

static uint32_t streamChunk( uint64_t total, uint64_t waves )
{
	uint64_t c = total / ( waves * 4 );
	c = ( c + 63 ) / 64 * 64;
#ifndef MVRT_CHUNK_MAX
#define MVRT_CHUNK_MAX 512 // rays per grab (1024 / 512 / 256: a launch alone 5885 / 6127 / 6152 Mrays/s -- finer grabs balance its end; pipelined job unchanged)
#endif
	if( c < 64 ) c = 64;
	if( c > MVRT_CHUNK_MAX ) c = MVRT_CHUNK_MAX;
	return (uint32_t)c;
}
static int streamGrid( uint64_t total, int nCUs )
{
	uint64_t need = ( total + 63 ) / 64;
	uint64_t cap = (uint64_t)nCUs * STREAM_WAVES_PER_CU;
	if( need < 1 ) need = 1;
	return (int)( need < cap ? need : cap );
}
// launches whose ray count the host knows (batch API, primary cast): the small-launch rule of the path tracer (kPtTraceStream) applied to the
// grid itself -- total / (64 * 8) waves, at least two per SIMD (primary cast 1920x1080 at 1024^3: 0.644 -> 0.46 ms)
static int streamGridKnown( uint64_t total, int nCUs )
{
	const uint64_t full = (uint64_t)streamGrid( total, nCUs );
#ifndef MVRT_KNOWN_RPL
#define MVRT_KNOWN_RPL 8
#endif
#ifndef MVRT_KNOWN_MINPCU
#define MVRT_KNOWN_MINPCU 8
#endif
	uint64_t want = total / ( 64ull * MVRT_KNOWN_RPL ) + 1;
	if( want < (uint64_t)nCUs * MVRT_KNOWN_MINPCU ) want = (uint64_t)nCUs * MVRT_KNOWN_MINPCU;
	return (int)( want < full ? want : full );
}
// lane capacity of a traversal workspace = row stride of the spill rows; rounded up to a power of two so that the kernel
// addresses row L of a lane with one shift-add on a 32-bit offset (traverse_stream.h)
uint64_t traceWorkspaceLanes()
{
	uint64_t lanes = (uint64_t)numCUs() * STREAM_WAVES_PER_CU * 64, p = 64;
	while( p < lanes ) p <<= 1;
	return p;
}

// =====================================================================================================
// mvrt_trace_batch: host-callable batch form of IntersectorOctreeGPU::intersect
// =====================================================================================================
static int persistentGrid( uint64_t items, int block, int numCUs, int blocksPerCU )
{
	uint64_t need = ( items + block - 1 ) / block;
	uint64_t cap = (uint64_t)numCUs * blocksPerCU;
	if( need < 1 ) need = 1;
	return (int)( need < cap ? need : cap );
}

static int g_numCUs = 0;
static int numCUs()
{
	if( g_numCUs == 0 )
	{
		int dev = 0;
		hipDeviceProp_t p;
		if( hipGetDevice( &dev ) == hipSuccess && hipGetDeviceProperties( &p, dev ) == hipSuccess ) g_numCUs = p.multiProcessorCount;
		if( g_numCUs <= 0 ) g_numCUs = 256;
	}
	return g_numCUs;
}

int launchTraceBatch( const SvoDev& svo, const TraceWorkspace& ws, uint64_t n, const float* rox, const float* roy, const float* roz, const float* rdx, const float* rdy,
					  const float* rdz, const uint8_t* isShadow, float* t, int32_t* nMajor, uint32_t* vIndex, uint32_t* descents, hipStream_t stream, const uint64_t* originPath )
{
	if( n == 0 ) return 0;
	if( ws.spill )
	{
		uint64_t* paths = vIndex ? ws.paths : nullptr;
		if( vIndex && ws.pathCap < n )
		{
			mvrtSetError( "internal: path scratch too small" );
			return 1;
		}
		BatchIO io = { rox, roy, roz, rdx, rdy, rdz, isShadow, t, nMajor, paths, descents };
		io.originPath = svo.embedded ? originPath : nullptr;
		io.levels = svo.levels;
		int grid = streamGridKnown( n, numCUs() );
		MVRT_HIP( hipMemsetAsync( ws.cursor, 0, 8, stream ) );
		if( svo.embedded ) hipLaunchKernelGGL( kTraceBatchStream<0>, dim3( grid ), dim3( 64 ), 0, stream, svo, n, io, ws, streamChunk( n, grid ) );
		else if( svo.tree ) hipLaunchKernelGGL( kTraceBatchStream<2>, dim3( grid ), dim3( 64 ), 0, stream, svo, n, io, ws, streamChunk( n, grid ) );
		else hipLaunchKernelGGL( kTraceBatchStream<1>, dim3( grid ), dim3( 64 ), 0, stream, svo, n, io, ws, streamChunk( n, grid ) );
		if( vIndex )
			hipLaunchKernelGGL( kResolveVIndex, dim3( persistentGrid( n, 256, numCUs(), 8 ) ), dim3( 256 ), 0, stream, svo, n, t, paths, isShadow, vIndex, (uchar4*)nullptr );
		MVRT_HIP( hipGetLastError() );
		return 0;
	}
	mvrtSetError( "internal: traversal workspace missing" );
	return 1;
}

// =====================================================================================================
// mvrt_render_primary: the `render` kernel (voxKernel.cu:437-483)
// =====================================================================================================
int launchRenderPrimary( const SvoDev& svo, const TraceWorkspace& ws, const CameraPinhole& cam, int W, int H, int showVertexColor, uchar4* rgba, float* t,
						 int32_t* nMajor, uint32_t* vIndex, uint32_t* descents, hipStream_t stream )
{
	uint64_t n = (uint64_t)W * H;
	if( n == 0 ) return 0;
	if( ws.spill )
	{
		PrimaryIO io;
		io.svo = svo;
		io.cam = cam;
		io.W = W;
		io.H = H;
		io.showVertexColor = showVertexColor;
		io.rgba = rgba;
		io.tOut = t;
		io.nMajorOut = nMajor;
		const bool needPaths = vIndex || ( showVertexColor && rgba );
		if( needPaths && ( ws.pathCap < n || !t ) )
		{
			mvrtSetError( "internal: path scratch too small" );
			return 1;
		}
		io.pathOut = needPaths ? ws.paths : nullptr;
		io.descentsOut = descents;
		io.rdKeep = mk3( 0, 0, 0 );
		int grid = streamGridKnown( n, numCUs() );
		MVRT_HIP( hipMemsetAsync( ws.cursor, 0, 8, stream ) );
		if( svo.embedded ) hipLaunchKernelGGL( kRenderPrimaryStream<0>, dim3( grid ), dim3( 64 ), 0, stream, io, ws, streamChunk( n, grid ) );
		else if( svo.tree ) hipLaunchKernelGGL( kRenderPrimaryStream<2>, dim3( grid ), dim3( 64 ), 0, stream, io, ws, streamChunk( n, grid ) );
		else hipLaunchKernelGGL( kRenderPrimaryStream<1>, dim3( grid ), dim3( 64 ), 0, stream, io, ws, streamChunk( n, grid ) );
		if( needPaths )
			hipLaunchKernelGGL( kResolveVIndex, dim3( persistentGrid( n, 256, numCUs(), 8 ) ), dim3( 256 ), 0, stream, svo, n, t, ws.paths, (const uint8_t*)nullptr, vIndex,
								showVertexColor ? rgba : (uchar4*)nullptr );
		MVRT_HIP( hipGetLastError() );
		return 0;
	}
	mvrtSetError( "internal: traversal workspace missing" );
	return 1;
}

// =====================================================================================================
// Stable compaction: count per 256-item virtual block -> single-workgroup exclusive scan -> rank
// =====================================================================================================
// number of set flags in each 256-item block
template <class Pred>
MVRT_DI void blockCountBody( Pred pred, uint64_t n, uint32_t* __restrict__ blockCount )
{
	__shared__ uint32_t waveCnt[CBLOCK / WAVE];
	const uint64_t nBlocks = ( n + CBLOCK - 1 ) / CBLOCK;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * CBLOCK + threadIdx.x;
		bool keep = i < n ? pred( i ) : false;
		unsigned long long b = __ballot( keep );
		if( ( threadIdx.x & ( WAVE - 1 ) ) == 0 ) waveCnt[threadIdx.x / WAVE] = __popcll( b );
		__syncthreads();
		if( threadIdx.x == 0 ) blockCount[vb] = waveCnt[0] + waveCnt[1] + waveCnt[2] + waveCnt[3];
		__syncthreads();
	}
}

// in-block stable rank of a kept item; every thread of the block must call it
MVRT_DI uint32_t blockRank( bool keep, uint32_t* waveCnt /* LDS, 4 entries */ )
{
	unsigned long long b = __ballot( keep );
	uint32_t lane = threadIdx.x & ( WAVE - 1 );
	uint32_t w = threadIdx.x / WAVE;
	uint32_t inWave = __popcll( b & ( ( 1ull << lane ) - 1ull ) );
	if( lane == 0 ) waveCnt[w] = __popcll( b );
	__syncthreads();
	uint32_t off = 0;
	for( uint32_t k = 0; k < w; k++ ) off += waveCnt[k];
	__syncthreads();
	return off + inWave;
}

// exclusive scan of blockCount[0..nBlocks) in place by ONE workgroup of 1024 threads; total -> *totalOut.
// nBlocks is read from device memory when nDev != null (n = *nDev items).  Tiles of 4096 counts: every thread owns four consecutive counts
// (one 16-byte load and store, coalesced), scans them, the workgroup scans the 1024 thread sums (wave shuffles + 16 wave totals in LDS) and a
// running total carries over to the next tile.  The count arrays are padded to a multiple of 4 entries (allocWorkSlot, mvrt_compact_indices).
template <int THREADS>
__global__ void __launch_bounds__( THREADS ) kScanBlockCounts( uint32_t* __restrict__ blockCount, uint64_t nItemsHost, const uint32_t* __restrict__ nItemsDev,
															 uint32_t* __restrict__ totalOut, unsigned long long* __restrict__ rayStats = nullptr, int nKinds = 0, int shadowKind = 0 )
{
	__shared__ uint32_t waveTot[THREADS / WAVE];
	__shared__ uint32_t carryLds;
	uint64_t nItems = nItemsDev ? (uint64_t)*nItemsDev : nItemsHost;
	if( rayStats && threadIdx.x == 0 ) // ray accounting of the stage that was just traced (intersect() calls)
	{
		atomicAdd( &rayStats[0], (unsigned long long)nItems * nKinds );
		if( shadowKind ) atomicAdd( &rayStats[1], (unsigned long long)nItems );
	}
	const uint32_t nBlocks = (uint32_t)( ( nItems + CBLOCK - 1 ) / CBLOCK );
	const uint32_t lane = threadIdx.x & ( WAVE - 1 ), wave = threadIdx.x / WAVE;
	uint32_t carry = 0;
	for( uint32_t base = 0; base < nBlocks; base += THREADS * 4 )
	{
		const uint32_t i0 = base + threadIdx.x * 4;
		uint4 c = make_uint4( 0u, 0u, 0u, 0u );
		if( i0 < nBlocks ) c = *(const uint4*)( blockCount + i0 ); // entries past nBlocks inside the last quad are padding: masked below
		if( i0 + 1 >= nBlocks ) c.y = 0u;
		if( i0 + 2 >= nBlocks ) c.z = 0u;
		if( i0 + 3 >= nBlocks ) c.w = 0u;
		const uint32_t mine = c.x + c.y + c.z + c.w;
		uint32_t incl = mine; // inclusive scan of the thread sums inside the wave
		for( int o = 1; o < WAVE; o <<= 1 )
		{
			const uint32_t v = __shfl_up( incl, o, WAVE );
			if( (int)lane >= o ) incl += v;
		}
		if( lane == WAVE - 1 ) waveTot[wave] = incl;
		__syncthreads();
		uint32_t before = carry;
		for( uint32_t w = 0; w < wave; w++ ) before += waveTot[w];
		if( threadIdx.x == THREADS - 1 ) carryLds = before + incl;
		const uint32_t e0 = before + incl - mine;
		if( i0 + 3 < nBlocks ) *(uint4*)( blockCount + i0 ) = make_uint4( e0, e0 + c.x, e0 + c.x + c.y, e0 + c.x + c.y + c.z );
		else if( i0 < nBlocks ) // the last, partial quad: nothing is written past nBlocks
		{
			blockCount[i0] = e0;
			if( i0 + 1 < nBlocks ) blockCount[i0 + 1] = e0 + c.x;
			if( i0 + 2 < nBlocks ) blockCount[i0 + 2] = e0 + c.x + c.y;
		}
		__syncthreads();
		carry = carryLds;
		__syncthreads();
	}
	if( threadIdx.x == 0 && totalOut ) *totalOut = carry;
}

__global__ void __launch_bounds__( CBLOCK ) kCountFlags( const uint8_t* __restrict__ keep, uint64_t n, uint32_t* __restrict__ blockCount )
{
	blockCountBody( [=]( uint64_t i ) { return keep[i] != 0; }, n, blockCount );
}
__global__ void __launch_bounds__( CBLOCK ) kRankFlags( const uint8_t* __restrict__ keep, uint64_t n, const uint32_t* __restrict__ blockOffset, uint32_t* __restrict__ dstIndex )
{
	__shared__ uint32_t waveCnt[CBLOCK / WAVE];
	const uint64_t nBlocks = ( n + CBLOCK - 1 ) / CBLOCK;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		uint64_t i = vb * CBLOCK + threadIdx.x;
		bool k = i < n ? keep[i] != 0 : false;
		uint32_t r = blockRank( k, waveCnt );
		if( i < n ) dstIndex[i] = k ? blockOffset[vb] + r : 0xFFFFFFFFu;
	}
}

int launchCompactIndices( const uint8_t* keep, uint64_t n, uint32_t* dstIndex, uint32_t* kept, uint32_t* blockScratch, hipStream_t stream )
{
	int grid = persistentGrid( n, CBLOCK, numCUs(), 8 );
	hipLaunchKernelGGL( kCountFlags, dim3( grid ), dim3( CBLOCK ), 0, stream, keep, n, blockScratch );
	hipLaunchKernelGGL( kScanBlockCounts<1024>, dim3( 1 ), dim3( 1024 ), 0, stream, blockScratch, n, (const uint32_t*)nullptr, kept, (unsigned long long*)nullptr, 0, 0 );
	hipLaunchKernelGGL( kRankFlags, dim3( grid ), dim3( CBLOCK ), 0, stream, keep, n, blockScratch, dstIndex );
	MVRT_HIP( hipGetLastError() );
	return 0;
}

// =====================================================================================================
// Wavefront path tracer
// =====================================================================================================
struct PtParams
{
	SvoDev svo;
	HdriDev hdri;
	const float2* pmj;
	CameraPinhole cams[MVRT_MAX_BATCH]; // one camera per merged step
	PtFrame frame;
	PtBuffers buf;
	int hdriEnabled;  // hdri.isEnabled(), renderCommon.hpp:467-470
	int extraSamples; // nSampleExtraDirect = hasEmission ? 1 : 0, voxKernel.cu:721
	int useHints;	  // secondary rays start below the root (traverse_stream.h); 0 = every ray from the root (A/B, tests)
};

// owned (local) pixel -> global pixel index.  Blocks of 256 pixels dealt round-robin over tiles.
MVRT_DI uint32_t globalPixel( const PtFrame& f, uint32_t localPixel )
{
	uint32_t lb = localPixel / MVRT_TILE_PIXELS;
	return ( lb * f.tileCount + f.tileIndex ) * MVRT_TILE_PIXELS + ( localPixel % MVRT_TILE_PIXELS );
}

// sample id ("task") of a batch of nSteps merged steps: ((step * validOwnedPixels + localPixel) * 16 + localSpp)
MVRT_DI void decodeTask( const PtFrame& f, uint32_t task, uint32_t* step, uint32_t* localPixel, uint32_t* localSpp )
{
	const uint32_t ps = task / MVRT_SPP_PER_STEP;
	*localSpp = task % MVRT_SPP_PER_STEP;
	const uint32_t vp = (uint32_t)f.validOwnedPixels;
	*step = ps / vp;
	*localPixel = ps - *step * vp;
}

// number of PMJ dimensions consumed before the shading of depth k (voxKernel.cu:662-666,699-700,724,741)
MVRT_DI int dimBase( int k, int hdriEnabled, int extraSamples ) { return 2 + k * ( ( hdriEnabled ? 2 : 0 ) + 1 ) + ( k > 0 ? extraSamples : 0 ); }

// ---- generate: camera samples -> primary rays (voxKernel.cu:635-667) -------------------------------
__global__ void __launch_bounds__( 256 ) kPtGenerate( PtParams P )
{
	const uint64_t n = P.frame.validOwnedPixels * MVRT_SPP_PER_STEP * P.frame.nSteps;
	const PathSet& o = P.buf.set[0];
	for( uint64_t task = (uint64_t)blockIdx.x * 256 + threadIdx.x; task < n; task += (uint64_t)gridDim.x * 256 )
	{
		uint32_t step, localPixel, localSpp;
		decodeTask( P.frame, (uint32_t)task, &step, &localPixel, &localSpp );
		uint32_t pixelIdx = globalPixel( P.frame, localPixel );
		int x = pixelIdx % P.frame.width;
		int y = pixelIdx / P.frame.width;
		uint32_t spp = ( P.frame.iteration + step ) * MVRT_SPP_PER_STEP + localSpp; // voxKernel.cu:642
		uint32_t stream = hashCombine2( 0u, pixelIdx );
		f2 camU = pmjSample2d( P.pmj, spp, 0, stream );
		f2 lensU = pmjSample2d( P.pmj, spp, 1, stream );
		f3 ro, rd;
		cameraShootThinLens( P.cams[step], &ro, &rd, x, y, camU.x, camU.y, P.frame.width, P.frame.height, lensU.x, lensU.y );
		o.task[task] = (uint32_t)task;
		o.rox[task] = ro.x;
		o.roy[task] = ro.y;
		o.roz[task] = ro.z;
		o.rdx[task] = rd.x;
		o.rdy[task] = rd.y;
		o.rdz[task] = rd.z;
	}
	for( uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < ( n + CBLOCK - 1 ) / CBLOCK; b += (uint64_t)gridDim.x * 256 ) P.buf.blockCount[b] = 0u;
	if( blockIdx.x == 0 && threadIdx.x == 0 )
	{
		P.buf.liveCount[0] = (uint32_t)n;
		for( int k = 0; k < 16; k++ ) P.buf.cursors[k] = 0ull; // per-stage ray cursors of the persistent traversal waves
		atomicAdd( &P.buf.stats[5], (unsigned long long)n );
	}
}

// ---- trace: all rays of one stage.  Ray r = kind * n + path; kind 0 bounce/primary, 1 shadow, 2 extra -
// Opaque view of a kernarg pointer: the asm makes the value unknown to the optimiser at this point, so the ~25 array
// pointers read through it are (re)loaded with scalar loads where they are used -- at refill time -- instead of being
// hoisted out of the traversal loop, where they would occupy ~60 SGPRs and push the node pointer out of registers.
// The table is written by the host before the launch and never by a kernel: it is read through the CONSTANT address space, so the
// pointer fetches are scalar loads that the compiler may batch freely (as plain global loads each of them had to be re-fetched, and
// waited for, after every result store that might alias the table: ~9 dependent memory round trips per refill instead of ~3).
template <class T>
MVRT_DI const __attribute__( ( address_space( 4 ) ) ) T* opaquePtr( const T* p )
{
	asm volatile( "" : "+s"( p ) );
	return (const __attribute__( ( address_space( 4 ) ) ) T*)p;
}

struct PtIO
{
	const PtBuffers* table; // device-resident pointer table (read through opaquePtr)
	int setIdx;
	uint32_t n;
	int shadowKind;
	int useHint; // stage > 0 on an embedded-mask octree: PathSet::org holds the start-below-the-root hint of every path
#ifdef MVRT_UTIL_STATS
	unsigned long long utilIters = 0, utilActive = 0, utilTailIters = 0, utilTailActive = 0, utilRefillClocks = 0, utilTotalClocks = 0;
	uint32_t utilMaxRayIters = 0;
#endif
	uint32_t dNormal, dShadow, nHits; // per-lane tallies (a lane sees far fewer than 2^32 descents per launch)
	MVRT_DI int kindOf( uint32_t r, uint32_t* i ) const
	{
		const uint32_t kindSlot = ( r >= n ? 1u : 0u ) + ( r >= 2u * n ? 1u : 0u );
		*i = r - ( kindSlot == 0u ? 0u : ( kindSlot == 1u ? n : 2u * n ) ); // (no integer multiply: quarter rate)
		return kindSlot == 0 ? 0 : ( kindSlot == 1 ? ( shadowKind ? 1 : 2 ) : 2 );
	}
	MVRT_DI bool load( uint32_t r, f3* ro, f3* rd, uint32_t* hint ) const
	{
		const __attribute__( ( address_space( 4 ) ) ) PtBuffers& buf = *opaquePtr( table );
		const __attribute__( ( address_space( 4 ) ) ) PathSet& in = buf.set[setIdx];
		uint32_t i;
		const int kind = kindOf( r, &i );
		*ro = mk3( in.rox[i], in.roy[i], in.roz[i] );
		if( useHint ) *hint = in.org[i]; // every ray of a path starts on the voxel the path hit last
		if( kind == 0 ) *rd = mk3( in.rdx[i], in.rdy[i], in.rdz[i] );
		else if( kind == 1 ) *rd = mk3( buf.sx[i], buf.sy[i], buf.sz[i] );
		else *rd = mk3( buf.ex[i], buf.ey[i], buf.ez[i] );
		return kind == 1;
	}
	MVRT_DI void store( uint32_t r, const StreamHit& h, bool )
	{
		const __attribute__( ( address_space( 4 ) ) ) PtBuffers& buf = *opaquePtr( table );
		uint32_t i;
		const int kind = kindOf( r, &i );
		const bool isHit = h.t != MVRT_MAXF;
		if( kind == 0 )
		{
			buf.hitT[i] = h.t;
			buf.hitPath[i] = h.path;
			buf.hitN[i] = (uint8_t)h.nMajor;
			if( isHit ) atomicAdd( &buf.blockCount[i / CBLOCK], 1u ); // survivors per 256-path block (order-free count)
		}
		else if( kind == 1 )
		{
			buf.hitS[i] = isHit ? 1 : 0;
		}
		else
		{
			buf.hitE[i] = isHit ? 1 : 0;
			buf.hitEPath[i] = h.path;
		}
		dShadow += kind == 1 ? h.descents : 0u;
		dNormal += kind == 1 ? 0u : h.descents;
		nHits += ( kind != 1 && isHit ) ? 1u : 0u;
	}
};
template <int FL>
__global__ void __launch_bounds__( 64, MVRT_WAVES_OF( FL ) ) kPtTraceStream( PtParams P, TraceWorkspace ws, int stage, int setIdx, int nKinds, int shadowKind, uint32_t chunk,
																			uint32_t raysPerLane, uint32_t minWaves )
{
	__shared__ uint4 ring[MVRT_RING_OF( FL ) * 64];
	__shared__ uint32_t ringMask[FL == 0 ? 1 : ( FL == 2 ? 2 : 1 ) * MVRT_RING_OF( FL ) * 64]; // (tree flavour: two mask words per stacked node)
	PtIO io;
	io.table = P.buf.selfDev;
	io.setIdx = setIdx;
	io.n = P.buf.liveCount[stage];
	io.shadowKind = shadowKind;
	io.useHint = ( FL == 0 && stage > 0 && P.useHints ) ? 1 : 0;
	io.dNormal = io.dShadow = io.nHits = 0;
	const uint64_t total = (uint64_t)io.n * nKinds;
	// The grid is sized on the host for the upper bound (live counts stay on the device).  A launch with few rays is not throughput- but
	// latency-bound -- its time is the longest ray's iterations times the time of one iteration, and an iteration takes as long as the
	// SIMD's co-resident waves make it -- so it keeps only as many waves as give each lane raysPerLane rays (at least minWaves): fewer waves
	// per SIMD, each with its lanes refilled from a longer stretch of the stream.  The other workgroups retire at once.
	uint64_t nWaves = gridDim.x;
	if( raysPerLane )
	{
		uint64_t wantWaves = total / ( 64ull * raysPerLane ) + 1;
		if( wantWaves < minWaves ) wantWaves = minWaves;
		if( blockIdx.x >= wantWaves ) return;
		if( wantWaves < nWaves ) nWaves = wantWaves;
	}
	uint64_t c = total / ( nWaves * 4 );
	c = ( c + 63 ) / 64 * 64;
	if( c < 64 ) c = 64;
	if( c > chunk ) c = chunk;
	traceStream<FL>( makeTraceCore( P.svo ), io, total, P.buf.cursors + stage, (uint32_t)c, ring, ws.spill, ws.spillStride, (uint64_t)blockIdx.x * 64 + threadIdx.x, ringMask,
						ws.spillMask, ws.spillMask2 );
	unsigned long long dN = waveSum( (unsigned long long)io.dNormal ), dS = waveSum( (unsigned long long)io.dShadow ), nH = waveSum( (unsigned long long)io.nHits );
#ifdef MVRT_UTIL_STATS
	if( threadIdx.x == 0 )
	{
		atomicAdd( &P.buf.stats[8 + 0], io.utilIters );
		atomicAdd( &P.buf.stats[8 + 1], io.utilActive );
		atomicAdd( &P.buf.stats[8 + 2], io.utilTailIters );
		atomicAdd( &P.buf.stats[8 + 3], io.utilTailActive );
		atomicAdd( &P.buf.stats[12], io.utilRefillClocks );
		atomicAdd( &P.buf.stats[13], io.utilTotalClocks );
		atomicMax( &P.buf.stats[16 + stage], io.utilTailIters ); // longest-lived wave of the stage's launches, in node-visit iterations
		atomicAdd( &P.buf.stats[32 + stage], 1ull );			  // waves that worked
	}
	{
		unsigned long long m = io.utilMaxRayIters;
		for( int o = 32; o > 0; o >>= 1 )
		{
			const unsigned long long t = __shfl_xor( m, o );
			m = t > m ? t : m;
		}
		if( threadIdx.x == 0 ) atomicMax( &P.buf.stats[48 + stage], m ); // longest ray of the stage, in node-visit iterations
	}
#endif
	if( threadIdx.x == 0 )
	{
		if( dN ) atomicAdd( &P.buf.stats[2], dN );
		if( dS ) atomicAdd( &P.buf.stats[3], dS );
		if( nH ) atomicAdd( &P.buf.stats[4], nH );
	}
}

// ---- HDRI lookups (renderCommon.hpp:175-180,354-365,367-465) ---------------------------------------
MVRT_DI f3 hdriSampleNearest( const HdriDev& h, f3 d, bool isPrimary )
{
	int w = isPrimary ? h.widthPrimary : h.width;
	int hh = isPrimary ? h.heightPrimary : h.height;
	float phi = mvrt_atan2( d.z, d.x ) + MVRT_PI;
	float theta = mvrt_atan2( sqrtf( d.x * d.x + d.z * d.z ), d.y );
	float u = phi / ( MVRT_PI * 2.0f );
	float v = theta / MVRT_PI;
	int x = (int)smin( smax( u * w, 0.0f ), (float)( w - 1.0f ) );
	int y = (int)smin( smax( v * hh, 0.0f ), (float)( hh - 1.0f ) );
	uint64_t index = (uint64_t)y * w + x;
	const float4* px = ( isPrimary && h.pixelsPrimary ) ? h.pixelsPrimary : h.pixels;
	// no map loaded: only possible with the lighting switched off (scale 0: mvrt_pt_step refuses scale > 0 without a map), where a loaded map would
	// contribute pixel * 0 -- the reference reads m_pixels unconditionally here (voxKernel.cu:682) and its applications always load one
	if( !px ) return mk3( 0.0f, 0.0f, 0.0f );
	float4 c = px[index];
	return mk3( c.x, c.y, c.z ) * h.scale;
}
MVRT_DI uint32_t satH( const HdriDev& h, const uint32_t* s, uint32_t x ) { return x == 0 ? 0u : s[h.width * ( h.height - 1 ) + x - 1]; }
MVRT_DI uint32_t satV( const HdriDev& h, const uint32_t* s, uint32_t x, uint32_t y )
{
	if( y == 0 ) return 0u;
	uint32_t s0 = x == 0 ? 0u : s[h.width * ( y - 1 ) + ( x - 1 )];
	uint32_t s1 = s[h.width * ( y - 1 ) + x];
	return s1 - s0;
}
MVRT_DI uint32_t satCount( const HdriDev& h, const uint32_t* s, uint32_t x, uint32_t y )
{
	uint32_t a = ( x == 0 || y == 0 ) ? 0u : s[h.width * ( y - 1 ) + ( x - 1 )];
	uint32_t b = ( y == 0 ) ? 0u : s[h.width * ( y - 1 ) + x];
	uint32_t c = ( x == 0 ) ? 0u : s[h.width * y + ( x - 1 )];
	uint32_t d = s[h.width * y + x];
	return ( d - b ) + ( a - c );
}
MVRT_DI void hdriImportanceSample( const HdriDev& h, f3* dir, f3* L, float* pdf, f3 N, float u0, float u1, float u2, float u3 )
{
	const uint32_t* s = h.sat;
	const float k = 0.8f;
	if( k < N.x ) s = h.sats[0];
	else if( N.x < -k ) s = h.sats[1];
	else if( k < N.y ) s = h.sats[2];
	else if( N.y < -k ) s = h.sats[3];
	else if( k < N.z ) s = h.sats[4];
	else if( N.z < -k ) s = h.sats[5];

	int i = 0, j = h.width; // upper_bound_f, renderCommon.hpp:182-202
	while( i < j )
	{
		int m = ( i + j ) / 2;
		float value = (float)satH( h, s, m ) / 4294967296.0f; // (float)0xFFFFFFFFu
		if( value <= u0 ) i = m + 1;
		else j = m;
	}
	uint32_t X = (uint32_t)( i - 1 );
	uint32_t vol = satH( h, s, X + 1 ) - satH( h, s, X );
	i = 0;
	j = h.height;
	while( i < j )
	{
		int m = ( i + j ) / 2;
		float value = (float)satV( h, s, X, m ) / (float)vol;
		if( value <= u1 ) i = m + 1;
		else j = m;
	}
	uint32_t Y = (uint32_t)( i - 1 );
	float pSelection = (float)satCount( h, s, X, Y ) / 4294967296.0f;
	float dTheta = MVRT_PI / (float)h.height;
	float dPhi = 2.0f * MVRT_PI / (float)h.width;
	float theta = Y * dTheta;
	float dH = 2.0f * mvrt_sin( dTheta * 0.5f ) * mvrt_sin( dTheta * 0.5f + theta );
	float sr = dH * dPhi;
	float sY = mixf( mvrt_cos( theta ), mvrt_cos( theta + dTheta ), u2 );
	float phi = dPhi * ( (float)X + u3 ) + MVRT_PI;
	float sX, sZ;
	mvrt_sincos( phi, &sZ, &sX );
	float sinTheta = sqrtf( smax( 1.0f - sY * sY, 0.0f ) );
	*dir = mk3( sX * sinTheta, sY, sZ * sinTheta );
	*pdf = pSelection / sr;
	float4 color = h.pixels[Y * h.width + X];
	*L = mk3( color.x, color.y, color.z ) * h.scale;
}

// sampleLambertian + GetOrthonormalBasis, renderCommon.hpp:119-151
MVRT_DI f3 sampleLambertian( float a, float b, f3 Ng )
{
	float r = sqrtf( a );
	float theta = b * MVRT_PI * 2.0f;
	float sn, cs;
	mvrt_sincos( theta, &sn, &cs );
	float x = r * cs;
	float y = r * sn;
	float z = sqrtf( smax( 1.0f - a, 0.0f ) );
	const float sign = copysignf( 1.0f, Ng.z );
	const float aa = -1.0f / ( sign + Ng.z );
	const float bb = Ng.x * Ng.y * aa;
	f3 xaxis = mk3( 1.0f + sign * Ng.x * Ng.x * aa, sign * bb, -sign * Ng.x );
	f3 yaxis = mk3( bb, sign + Ng.y * Ng.y * aa, -Ng.y );
	return xaxis * x + yaxis * y + Ng * z;
}

MVRT_DI f3 voxelEmission( const SvoDev& s, uint32_t vIndex, bool withScale ) // IntersectorOctreeGPU.hpp:256-259
{
	return rawReflectance( s.attrs[vIndex].y ) * ( withScale ? s.emissionScale : 1.0f );
}

// ---- shade + compact: stage k consumes the hits of trace k, adds the contributions that became known,
// and for paths whose bounce ray hit, shades the hit (depth k of the reference loop, voxKernel.cu:691-760),
// writing the survivor to its STABLE compacted slot of the other PathSet. --------------------------------
#ifndef MVRT_SHADE_WAVES
#define MVRT_SHADE_WAVES 5 // waves per SIMD the shade kernel is register-budgeted for
#endif
template <int WAVES> // register budget: waves per SIMD the kernel is compiled for
__global__ void __launch_bounds__( CBLOCK, WAVES ) kPtShade( PtParams P, int stage, int inSet )
{
	__shared__ uint32_t waveCnt[CBLOCK / WAVE];
	const uint64_t n = P.buf.liveCount[stage];
	const PathSet& in = P.buf.set[inSet];
	const PathSet& out = P.buf.set[inSet ^ 1];
	const bool lastStage = stage >= MVRT_MAX_DEPTH;
	const uint64_t nBlocks = ( n + CBLOCK - 1 ) / CBLOCK;
	for( uint64_t vb = blockIdx.x; vb < nBlocks; vb += gridDim.x )
	{
		const uint64_t i = vb * CBLOCK + threadIdx.x;
		const bool valid = i < n;
		bool alive = false;
		uint32_t task = 0, vIndex = 0;
		uint2 attr = make_uint2( 0u, 0u );
		uint64_t hitPath = 0;
		int nMajor = 0;
		float t = MVRT_MAXF;
		f3 ro, rd, T, L;
		if( valid )
		{
			task = in.task[i];
			ro = mk3( in.rox[i], in.roy[i], in.roz[i] );
			rd = mk3( in.rdx[i], in.rdy[i], in.rdz[i] );
			t = P.buf.hitT[i];
			nMajor = P.buf.hitN[i];
			const bool hit = t != MVRT_MAXF;
			// the persistent traversal reports the hit voxel's path; its index is summed here, all lanes together
			hitPath = hit ? P.buf.hitPath[i] : 0ull;
			vIndex = hit ? voxelIndexFromPath( P.svo, hitPath ) : 0u;
			if( hit ) attr = P.svo.attrs[vIndex]; // colour and emission of the hit voxel: one 8-byte gather serves both uses below
			if( stage == 0 )
			{
				T = mk3( 1.0f, 1.0f, 1.0f );
				L = mk3( 0.0f, 0.0f, 0.0f );
				if( !hit ) // voxKernel.cu:678-684
				{
					L = L + T * hdriSampleNearest( P.hdri, rd, true );
				}
				else // :685-689
				{
					L = L + T * rawReflectance( attr.y ); // (IntersectorOctreeGPU.hpp:256-259 without the scale)
				}
			}
			else
			{
				T = mk3( in.Tx[i], in.Ty[i], in.Tz[i] );
				L = mk3( in.Lx[i], in.Ly[i], in.Lz[i] );
				if( P.hdriEnabled && P.buf.hitS[i] == 0 ) // :712-715, contribution computed when the ray was generated
				{
					L = L + mk3( in.nx[i], in.ny[i], in.nz[i] );
				}
				if( stage == 1 && P.extraSamples ) // :722-738
				{
					if( P.buf.hitE[i] )
					{
						const uint32_t ev = voxelIndexFromPath( P.svo, P.buf.hitEPath[i] );
						f3 Le = voxelEmission( P.svo, ev, true );
						L = L + T * Le / (float)( 1 + P.extraSamples );
					}
				}
				if( hit ) // :750-759
				{
					f3 Le = rawReflectance( attr.y ) * P.svo.emissionScale; // IntersectorOctreeGPU.hpp:256-259
					L = L + T * Le * ( stage == 1 ? 1.0f / (float)( 1 + P.extraSamples ) : 1.0f );
				}
			}
			alive = hit && !lastStage;
			if( !alive )
			{
				P.buf.Lsx[task] = L.x;
				P.buf.Lsy[task] = L.y;
				P.buf.Lsz[task] = L.z;
			}
		}
		if( lastStage ) continue; // nothing survives stage 8; uniform across the grid
		const uint32_t rank = blockRank( alive, waveCnt );
		const uint32_t blockOffset = P.buf.blockCount[vb];
		__syncthreads();
		if( threadIdx.x == 0 ) P.buf.blockCount[vb] = 0u; // the next stage's traversal counts its survivors into it again
		if( alive )
		{
			const uint64_t j = (uint64_t)blockOffset + rank;
			// depth = stage of the reference loop
			uint32_t step, localPixel, localSpp;
			decodeTask( P.frame, task, &step, &localPixel, &localSpp );
			const uint32_t spp = ( P.frame.iteration + step ) * MVRT_SPP_PER_STEP + localSpp;
			const uint32_t stream = hashCombine2( 0u, globalPixel( P.frame, localPixel ) );
			int dim = dimBase( stage, P.hdriEnabled, P.extraSamples );
			const f3 R = rawReflectance( attr.x ); // :693
			const f3 hitN = getHitN( nMajor, rd );				  // :694
			const f3 hitP = ro + rd * t;						  // :695
			f3 nee = mk3( 0.0f, 0.0f, 0.0f );
			if( P.hdriEnabled ) // :697-716
			{
				f2 u01 = pmjSample2d( P.pmj, spp, dim++, stream );
				f2 u23 = pmjSample2d( P.pmj, spp, dim++, stream );
				f3 sdir, emissive;
				float p;
				hdriImportanceSample( P.hdri, &sdir, &emissive, &p, hitN, u01.x, u01.y, u23.x, u23.y );
				nee = T * ( R / MVRT_PI ) * smax( dot3( hitN, sdir ), 0.0f ) * emissive / p;
				P.buf.sx[j] = sdir.x;
				P.buf.sy[j] = sdir.y;
				P.buf.sz[j] = sdir.z;
			}
			T = T * R;								// :718
			if( stage == 0 && P.extraSamples ) // :721-726
			{
				f2 u = pmjSample2d( P.pmj, spp, dim++, stream );
				f3 edir = sampleLambertian( u.x, u.y, hitN );
				P.buf.ex[j] = edir.x;
				P.buf.ey[j] = edir.y;
				P.buf.ez[j] = edir.z;
			}
			f2 u = pmjSample2d( P.pmj, spp, dim++, stream ); // :741-745
			f3 bdir = sampleLambertian( u.x, u.y, hitN );
			out.task[j] = task;
			if( P.svo.embedded ) out.org[j] = hintFromVoxelPath( hitPath, P.svo.levels ); // the rays of the next stage start on this voxel
			out.rox[j] = hitP.x;
			out.roy[j] = hitP.y;
			out.roz[j] = hitP.z;
			out.rdx[j] = bdir.x;
			out.rdy[j] = bdir.y;
			out.rdz[j] = bdir.z;
			out.Tx[j] = T.x;
			out.Ty[j] = T.y;
			out.Tz[j] = T.z;
			out.Lx[j] = L.x;
			out.Ly[j] = L.y;
			out.Lz[j] = L.z;
			out.nx[j] = nee.x;
			out.ny[j] = nee.y;
			out.nz[j] = nee.z;
		}
	}
}

// ---- accumulate: frameBuffer[p].xyz += sum of the 16 samples in ascending spp order, .w += 16 -------
// (voxKernel.cu:763-774; the reference's LDS atomicAdd order is nondeterministic, ascending is ours)
__global__ void __launch_bounds__( 256 ) kPtAccumulate( PtParams P, float4* __restrict__ fb )
{
	const uint64_t n = P.frame.validOwnedPixels;
	for( uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (uint64_t)gridDim.x * 256 )
	{
		float4 v = fb[p];
		for( int b = 0; b < P.frame.nSteps; b++ ) // merged steps are added one after the other, exactly as separate launches would
		{
			const uint64_t base = ( (uint64_t)b * n + p ) * MVRT_SPP_PER_STEP;
			const float4* sx = (const float4*)( P.buf.Lsx + base );
			const float4* sy = (const float4*)( P.buf.Lsy + base );
			const float4* sz = (const float4*)( P.buf.Lsz + base );
			float ax = 0.0f, ay = 0.0f, az = 0.0f;
#pragma unroll
			for( int q = 0; q < MVRT_SPP_PER_STEP / 4; q++ )
			{
				float4 vx = sx[q], vy = sy[q], vz = sz[q];
				ax += vx.x; ax += vx.y; ax += vx.z; ax += vx.w;
				ay += vy.x; ay += vy.y; ay += vy.z; ay += vy.w;
				az += vz.x; az += vz.y; az += vz.z; az += vz.w;
			}
			v.x += ax;
			v.y += ay;
			v.z += az;
			v.w += (float)MVRT_SPP_PER_STEP;
		}
		fb[p] = v;
	}
}

int launchPtStep( const SvoDev& svo, const TraceWorkspace& ws, const HdriDev& hdri, const float2* pmj, const CameraPinhole* cams, const PtFrame& frame,
				  const PtBuffers& buf, float4* frameBuffer, int nCUs, PtProfiler* prof, hipStream_t stream, hipEvent_t accumulateAfter )
{
	PtParams P;
	P.svo = svo;
	P.hdri = hdri;
	P.pmj = pmj;
	for( int b = 0; b < MVRT_MAX_BATCH; b++ ) P.cams[b] = cams[b < frame.nSteps ? b : 0];
	P.frame = frame;
	P.buf = buf;
	P.hdriEnabled = ( 0.0f < hdri.scale ) ? 1 : 0;
	P.extraSamples = svo.hasEmission ? 1 : 0;
	P.useHints = frame.useHints;
	if( nCUs <= 0 ) nCUs = numCUs();

	const uint64_t nSamples = frame.validOwnedPixels * MVRT_SPP_PER_STEP * frame.nSteps;
	if( nSamples == 0 ) return 0;
	if( nSamples * 3 >= 0xFFFFFFFFull )
	{
		mvrtSetError( "batch too large: %llu samples (ray indices are 32-bit)", (unsigned long long)nSamples );
		return 1;
	}
	if( nSamples > buf.cap )
	{
		mvrtSetError( "path buffers too small: %llu samples > capacity %llu", (unsigned long long)nSamples, (unsigned long long)buf.cap );
		return 1;
	}
	if( !ws.spill )
	{
		mvrtSetError( "internal: traversal workspace missing" );
		return 1;
	}

#define PROF_BEGIN( c ) if( prof ) prof->begin( c, stream )
#define PROF_END() if( prof ) prof->end( stream )

	PROF_BEGIN( MVRT_K_OTHER );
	hipLaunchKernelGGL( kPtGenerate, dim3( persistentGrid( nSamples, 256, nCUs, 8 ) ), dim3( 256 ), 0, stream, P );
	PROF_END();

	for( int stage = 0; stage <= MVRT_MAX_DEPTH; stage++ )
	{
		const int setIdx = stage & 1;
		const int shadowKind = ( stage > 0 && P.hdriEnabled ) ? 1 : 0;
		const int extraKind = ( stage == 1 && P.extraSamples ) ? 1 : 0;
		const int nKinds = 1 + shadowKind + extraKind;
		PROF_BEGIN( MVRT_K_TRACE );
		{
			int g = streamGrid( nSamples * nKinds, nCUs );
			uint32_t smallRpl = 0, smallMinW = 0;
			{
				// a pass that shares the GPU with a sibling pass on another stream takes only 1/div of the wave slots, so that the two
				// really run side by side (a full persistent grid would hold every slot until its own tail)
				static const int envDiv = (int)mvrtKnob( "MVRT_TRACE_GRID_DIV", 0 );
				// ... for its BIG launches only (the first bounces; live counts are on the device, so the stage stands in for them): a small
				// launch finishes sooner the more lanes it may use, and is gone before it can be in the sibling's way
				static const int divMaxStage = (int)mvrtKnob( "MVRT_TRACE_DIV_MAX_STAGE", 99 );
				const int div = stage > divMaxStage ? 1 : ( envDiv > 0 ? envDiv : frame.traceGridDiv );
				if( div > 1 && g > nCUs * STREAM_WAVES_PER_CU / div ) g = nCUs * STREAM_WAVES_PER_CU / div;
				// small launches keep only total / (64 * 16) waves alive, at least 2048 (two per SIMD): measured with the 8-slot ring on a 1/8 tile
				// share -6.5 % (dragon, rtcamp stand-in), full frame -1 %, closed scene unchanged (gpurun_out/sweep_small2.log)
				static const uint32_t envRpl = (uint32_t)mvrtKnob( "MVRT_SMALL_RPL", 16 );
				static const uint32_t envMinW = (uint32_t)mvrtKnob( "MVRT_SMALL_MINW", 2048 );
				smallRpl = envRpl;
				smallMinW = envMinW;
				// experiment knob: waves per CU of a full-grid traversal launch (32 = every slot the register budget allows and then some)
				static const int wpc = (int)mvrtKnob( "MVRT_TRACE_WAVES_PER_CU", 0 );
				if( wpc > 0 && g > nCUs * wpc / ( div > 1 ? div : 1 ) ) g = nCUs * wpc / ( div > 1 ? div : 1 );
			}
			if( svo.embedded )
				hipLaunchKernelGGL( kPtTraceStream<0>, dim3( g ), dim3( 64 ), 0, stream, P, ws, stage, setIdx, nKinds, shadowKind, streamChunk( nSamples * nKinds, g ), smallRpl, smallMinW );
			else if( svo.tree )
				hipLaunchKernelGGL( kPtTraceStream<2>, dim3( g ), dim3( 64 ), 0, stream, P, ws, stage, setIdx, nKinds, shadowKind, streamChunk( nSamples * nKinds, g ), smallRpl, smallMinW );
			else
				hipLaunchKernelGGL( kPtTraceStream<1>, dim3( g ), dim3( 64 ), 0, stream, P, ws, stage, setIdx, nKinds, shadowKind, streamChunk( nSamples * nKinds, g ), smallRpl, smallMinW );
		}
		PROF_END();
		PROF_BEGIN( MVRT_K_OTHER );
		// survivors per block were counted by the traversal's result stores; scan them (and account the rays)
		// (a 256-thread workgroup: a 1024-thread one needs all 16 wave slots of one CU at once and waits -- up to a millisecond, measured on a tile share -- while the
		// traversal waves of a sibling pass hold them)
		static const int scanThreads = (int)mvrtKnob( "MVRT_SCAN_THREADS", 256 );
		if( scanThreads == 1024 )
			hipLaunchKernelGGL( kScanBlockCounts<1024>, dim3( 1 ), dim3( 1024 ), 0, stream, buf.blockCount, (uint64_t)0, (const uint32_t*)( buf.liveCount + stage ),
								stage < MVRT_MAX_DEPTH ? buf.liveCount + stage + 1 : (uint32_t*)nullptr, buf.stats, nKinds, shadowKind );
		else
			hipLaunchKernelGGL( kScanBlockCounts<256>, dim3( 1 ), dim3( 256 ), 0, stream, buf.blockCount, (uint64_t)0, (const uint32_t*)( buf.liveCount + stage ),
								stage < MVRT_MAX_DEPTH ? buf.liveCount + stage + 1 : (uint32_t*)nullptr, buf.stats, nKinds, shadowKind );
		PROF_END();
		PROF_BEGIN( MVRT_K_SHADE );
		{
			// Two builds of the kernel: stages 0 and 1 -- where most paths END (69 % of the primaries of an open scene miss) and the work per path is loads and a few adds --
			// run the 8-waves-per-SIMD build (64 VGPRs, 44 bytes of scratch that the dying paths never touch); the deeper stages, whose paths mostly survive and run
			// the ~2000-instruction shading, the 5-wave build.  r03, serial-mode shade time per 4 steps, stages {} / {0} / {0,1} / {1..8} / all on the 8-wave build:
			// dragon 16.6 / 15.8 / 14.9 / 15.6 / 14.6 ms, closed cave 133.7 / 132.5 / 133.3 / 137.6 / 136.8 (tools/sweep_shade.sh)
			static const int dense = (int)mvrtKnob( "MVRT_SHADE8_STAGES", 3 );
			// The grid-stride grids hold exactly the workgroups that are RESIDENT at once (8 / 5 per CU): r02 launched 8 per CU for the 5-wave build too, and the three
			// that had to wait for a slot ran as a second, thin round -- most of what the 8-wave build seemed to gain.  5-wave build with 8 / 6 / 5 workgroups per CU:
			// dragon 16.6 / 14.1 / 14.0 ms of shade time per 4 serial steps, cave 133.7 / 132.3 / 129.2
			static const int bpc8 = (int)mvrtKnob( "MVRT_SHADE_BPC8", 8 ), bpc5 = (int)mvrtKnob( "MVRT_SHADE_BPC5", 5 );
			if( ( dense >> stage ) & 1 )
				hipLaunchKernelGGL( kPtShade<8>, dim3( persistentGrid( nSamples, CBLOCK, nCUs, bpc8 ) ), dim3( CBLOCK ), 0, stream, P, stage, setIdx );
			else
				hipLaunchKernelGGL( kPtShade<MVRT_SHADE_WAVES>, dim3( persistentGrid( nSamples, CBLOCK, nCUs, bpc5 ) ), dim3( CBLOCK ), 0, stream, P, stage, setIdx );
		}
		PROF_END();
		if( buf.dbgTasks && stage < MVRT_MAX_DEPTH ) // debug capture: the compacted survivor list this stage wrote (parity of the compaction indices)
			MVRT_HIP( hipMemcpyAsync( buf.dbgTasks + (uint64_t)stage * buf.cap, buf.set[setIdx ^ 1].task, nSamples * 4, hipMemcpyDeviceToDevice, stream ) );
	}
	// frame-buffer additions must happen in step order (fixed fp32 summation order): wait for the previous step's
	if( accumulateAfter ) MVRT_HIP( hipStreamWaitEvent( stream, accumulateAfter, 0 ) );
	PROF_BEGIN( MVRT_K_OTHER );
	hipLaunchKernelGGL( kPtAccumulate, dim3( persistentGrid( frame.validOwnedPixels, 256, nCUs, 8 ) ), dim3( 256 ), 0, stream, P, frameBuffer );
	PROF_END();
	MVRT_HIP( hipGetLastError() );
	return 0;
}
