// api.hip -- the C-ABI of libmvrt_hip.so (include/mvrt.h): host objects that own device memory and
// launch the kernels.  Mirrors the reference's host structs IntersectorOctreeGPU
// (IntersectorOctreeGPU.hpp:21-275) and PathTracer (PathTracer.hpp:14-170) behind opaque handles.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mvrt.h"
#include "launch.h"
#include "traverse_stream.h" // (hint geometry: hintTabLevelsOf, prefixTabEntries)

#define MVRT_EXPORT extern "C" __attribute__( ( visibility( "default" ) ) )

static thread_local char g_err[1024] = "";
void mvrtSetError( const char* fmt, ... )
{
	va_list ap;
	va_start( ap, fmt );
	vsnprintf( g_err, sizeof( g_err ), fmt, ap );
	va_end( ap );
}
MVRT_EXPORT const char* mvrt_last_error( void ) { return g_err; }

#define REQUIRE( cond, ... )          \
	do                                \
	{                                 \
		if( !( cond ) )               \
		{                             \
			mvrtSetError( __VA_ARGS__ ); \
			return 1;                 \
		}                             \
	} while( 0 )

// ---- runtime ------------------------------------------------------------------------------------------
MVRT_EXPORT int mvrt_device_count( int* count )
{
	MVRT_HIP( hipGetDeviceCount( count ) );
	return 0;
}
MVRT_EXPORT int mvrt_set_device( int device )
{
	MVRT_HIP( hipSetDevice( device ) );
	return 0;
}
MVRT_EXPORT int mvrt_device_name( char* buf, int bufLen )
{
	int dev = 0;
	MVRT_HIP( hipGetDevice( &dev ) );
	hipDeviceProp_t p;
	MVRT_HIP( hipGetDeviceProperties( &p, dev ) );
	snprintf( buf, bufLen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount );
	return 0;
}
MVRT_EXPORT int mvrt_stream_create( void** stream )
{
	hipStream_t s;
	MVRT_HIP( hipStreamCreate( &s ) );
	*stream = (void*)s;
	return 0;
}
MVRT_EXPORT int mvrt_stream_destroy( void* stream )
{
	MVRT_HIP( hipStreamDestroy( (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_stream_synchronize( void* stream )
{
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_device_synchronize( void )
{
	MVRT_HIP( hipDeviceSynchronize() );
	return 0;
}
MVRT_EXPORT int mvrt_malloc( void** dev, uint64_t bytes )
{
	MVRT_HIP( hipMalloc( dev, bytes ? bytes : 1 ) );
	return 0;
}
MVRT_EXPORT int mvrt_free( void* dev )
{
	if( dev ) MVRT_HIP( hipFree( dev ) );
	return 0;
}
MVRT_EXPORT int mvrt_memcpy_h2d( void* dev, const void* host, uint64_t bytes, void* stream )
{
	MVRT_HIP( hipMemcpyAsync( dev, host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream ) );
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_memcpy_d2d( void* dstDev, const void* srcDev, uint64_t bytes, void* stream )
{
	MVRT_HIP( hipMemcpyAsync( dstDev, srcDev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_memcpy_d2h( void* host, const void* dev, uint64_t bytes, void* stream )
{
	MVRT_HIP( hipMemcpyAsync( host, dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream ) );
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	return 0;
}

// small RAII device buffer (role of hipUtil.hpp:48-74 Buffer)
struct DevBuf
{
	void* p = nullptr;
	uint64_t bytes = 0;
	int alloc( uint64_t b )
	{
		release();
		bytes = b;
		MVRT_HIP( hipMalloc( &p, b ? b : 1 ) );
		return 0;
	}
	void release()
	{
		if( p ) (void)hipFree( p );
		p = nullptr;
		bytes = 0;
	}
	~DevBuf() { release(); }
	DevBuf() {}
	DevBuf( const DevBuf& ) = delete;
	void operator=( const DevBuf& ) = delete;
	template <class T> T* as() const { return (T*)p; }
};

// ---- IntersectorOctreeGPU -------------------------------------------------------------------------------
struct mvrt_svo
{
	Node64* nodes = nullptr;
	uint8_t* masks = nullptr;
	uint32_t* psumCold = nullptr; // non-embedded flavour only
	// tree flavour (GPU-built octrees without node sharing whose masks are not embedded): `nodes` = two-level bricks, `masks` = per-node mask,
	// treeFirst = per-node first child (reference numbering), node ranges per builder level
	uint32_t* treeFirst = nullptr;
	uint32_t tree = 0, treeRoot = 0, nBricks = 0;
	uint32_t treeLevelBase[24] = { 0 }, treeLevelCount[24] = { 0 };
	uint2* attrs = nullptr;
	uint64_t* morton = nullptr; // only after build()
	DevBuf kids;				// embedded flavour: children[8] per node, 32 B per node (what the traversal reads)
	DevBuf topTable;			// per-prefix start of the nVoxelsPSum walk (SvoDev::topTable), embedded flavour
	uint32_t topLevels = 0;
	DevBuf cellBlocks, cellEntries; // SvoDev::cellBlocks / cellEntries, only after build()
	uint32_t cellBits = 0;
	mutable DevBuf wsBuf;		// traversal workspace (spill rows + cursor), sized on demand
	mutable DevBuf pathBuf;
	mutable TraceWorkspace ws = { nullptr, 0, nullptr, nullptr, 0, nullptr };
	mvrt_svo_info info;
	uint8_t rootMask = 0;
	uint32_t leafPsumIsPopcount = 1; // (uploads: checked, see launchCheckLeafPsum)
	mvrt_pt* owner = nullptr; // the PathTracer this is the m_intersectorOctreeGPU of (its deferred / in-flight steps read this octree)
	mvrt_svo()
	{
		memset( &info, 0, sizeof( info ) );
		info.emissionScale = 7.5f; // IntersectorOctreeGPU.hpp:273
	}
	void cleanUp() // :26-38
	{
		if( nodes ) (void)hipFree( nodes );
		if( masks ) (void)hipFree( masks );
		if( psumCold ) (void)hipFree( psumCold );
		if( treeFirst ) (void)hipFree( treeFirst );
		treeFirst = nullptr;
		tree = treeRoot = nBricks = 0;
		if( attrs ) (void)hipFree( attrs );
		if( morton ) (void)hipFree( morton );
		nodes = nullptr;
		masks = nullptr;
		psumCold = nullptr;
		attrs = nullptr;
		morton = nullptr;
		topTable.release();
		kids.release();
		topLevels = 0;
		cellBlocks.release();
		cellEntries.release();
		cellBits = 0;
		float es = info.emissionScale;
		memset( &info, 0, sizeof( info ) );
		info.emissionScale = es;
	}
	int ensureWorkspace( uint64_t nPaths = 0 ) const // one per handle; users of one handle must be stream-ordered
	{
		if( nPaths > ws.pathCap )
		{
			if( pathBuf.alloc( nPaths * 12 ) ) return 1; // 8-byte path + 4-byte t scratch per ray
			ws.paths = pathBuf.as<uint64_t>();
			ws.pathCap = nPaths;
		}
		const uint64_t lanes = traceWorkspaceLanes();
		const uint64_t rows = 2 * (uint64_t)( info.levels ? info.levels : 1 ) + 2; // fast path: 1 row per level; irregular rays: 2 per slot
		const uint64_t bytes = 256 + rows * lanes * ( sizeof( uint4 ) + 2 * sizeof( uint32_t ) );
		if( wsBuf.bytes < bytes )
		{
			if( wsBuf.alloc( bytes ) ) return 1;
		}
		ws.cursor = (unsigned long long*)wsBuf.p;
		ws.spill = (uint4*)( (uint8_t*)wsBuf.p + 256 );
		ws.spillStride = lanes;
		ws.spillMask = (uint32_t*)( ws.spill + rows * lanes );
		ws.spillMask2 = ws.spillMask + rows * lanes;
		return 0;
	}
	SvoDev dev() const
	{
		SvoDev d;
		d.nodes = nodes;
		d.masks = masks;
		d.psumCold = psumCold;
		d.attrs = attrs;
		d.nNodes = info.numberOfNodes;
		d.nVoxels = info.numberOfVoxels;
		d.lower = mk3( info.lower[0], info.lower[1], info.lower[2] );
		d.upper = mk3( info.upper[0], info.upper[1], info.upper[2] );
		d.dps = info.dps;
		d.emissionScale = info.emissionScale;
		d.hasEmission = info.hasEmission;
		d.embedded = info.embeddedMask;
		d.levels = info.levels;
		d.rootIndex = info.numberOfNodes - 1; // root = last node, :250
		d.rootMask = rootMask;
		d.kids = kids.as<uint32_t>();
		d.topTable = topTable.as<uint2>();
		d.topLevels = topLevels;
		d.cellBlocks = cellEntries.p ? cellBlocks.as<uint32_t>() : nullptr;
		d.cellEntries = cellEntries.as<uint2>();
		d.cellBits = cellBits;
		d.tree = tree;
		d.treeRoot = treeRoot;
		d.leafPsumIsPopcount = leafPsumIsPopcount;
		return d;
	}
};

// The reference's step() passes m_intersectorOctreeGPU and m_hdri to the kernel BY VALUE at call time (PathTracer.hpp:150-169).  step() is
// deferred here, so every change of state a pending or in-flight step reads is preceded by launching (flush) or finishing (drain) those steps.
static int ptFlush( mvrt_pt* pt );
static int ptDrain( mvrt_pt* pt );
static int ownerFlush( const mvrt_svo* s ) { return s && s->owner ? ptFlush( s->owner ) : 0; }
static int ownerDrain( const mvrt_svo* s ) { return s && s->owner ? ptDrain( s->owner ) : 0; }

// after nodes are in place: the prefix table that shortens every nVoxelsPSum walk (voxelIndexFromPath).  7 levels = 16 MiB (measured: shade kernel 19.3 ms without, 17.2 ms with 6 levels, 16.6 ms with 7, 15.9 ms with 8 = 128 MiB, per 4 steps).
static int buildTopTable( mvrt_svo* s, hipStream_t st )
{
	s->topTable.release();
	s->kids.release();
	s->topLevels = 0;
	if( !s->info.embeddedMask || !s->nodes || s->info.levels == 0 ) return 0;
	// children array (32 B per node) + behind it the prefix tables of the start below the root (traverse_stream.h): one buffer, one base register
	if( s->kids.alloc( (uint64_t)s->info.numberOfNodes * 32 + prefixTabEntries( s->info.levels ) * 4 ) ) return 1;
	if( launchCopyKids( s->nodes, s->info.numberOfNodes, s->kids.as<uint32_t>(), st ) ) return 1;
	if( launchBuildPrefixRefs( s->kids.as<uint32_t>(), ( s->info.numberOfNodes - 1 ) | ( (uint32_t)s->rootMask << 24 ), hintTabLevelsOf( s->info.levels ),
							   s->kids.as<uint32_t>() + (uint64_t)s->info.numberOfNodes * 8, st ) )
		return 1;
	static const int envK = (int)mvrtKnob( "MVRT_TOP_LEVELS", 7 );
	uint32_t k = (uint32_t)( envK < 0 ? 0 : ( envK > 8 ? 8 : envK ) );
	if( k > s->info.levels ) k = s->info.levels;
	if( k == 0 ) return 0;
	if( s->topTable.alloc( ( 1ull << ( 3 * k ) ) * sizeof( uint2 ) ) ) return 1;
	if( launchBuildTopTable( s->nodes, s->info.numberOfNodes - 1, k, s->topTable.as<uint2>(), st ) ) return 1;
	s->topLevels = k;
	return 0;
}

static int ilog2Exact( int v )
{
	int l = 0;
	while( ( 1 << l ) < v ) l++;
	return ( 1 << l ) == v ? l : -1;
}
static void setBounds( mvrt_svo* s, const float origin[3], float dps, int gridRes )
{
	// IntersectorOctreeGPU.hpp:78-80: m_upper = origin + float3{dps,dps,dps} * (float)gridRes
	for( int k = 0; k < 3; k++ )
	{
		s->info.lower[k] = origin[k];
		s->info.upper[k] = origin[k] + dps * (float)gridRes;
	}
	s->info.dps = dps;
	s->info.gridRes = gridRes;
	s->info.levels = ilog2Exact( gridRes );
}

MVRT_EXPORT int mvrt_svo_create( mvrt_svo** out )
{
	*out = new mvrt_svo();
	return 0;
}
MVRT_EXPORT int mvrt_svo_destroy( mvrt_svo* svo )
{
	if( svo )
	{
		svo->cleanUp();
		delete svo;
	}
	return 0;
}

MVRT_EXPORT int mvrt_svo_upload( mvrt_svo* svo, const void* nodes68Host, uint32_t numberOfNodes, const void* attribs8Host, uint32_t numberOfVoxels, const float origin[3],
								 float dps, int gridRes, int hasEmission, int embeddedMask, void* stream )
{
	REQUIRE( svo && nodes68Host && numberOfNodes > 0, "mvrt_svo_upload: empty octree" );
	REQUIRE( attribs8Host || numberOfVoxels == 0, "mvrt_svo_upload: %u voxels without attributes", numberOfVoxels );
	REQUIRE( ilog2Exact( gridRes ) > 0, "gridRes %d is not a power of two >= 2 (IntersectorOctreeGPU.hpp:48-51)", gridRes );
	REQUIRE( !embeddedMask || numberOfNodes < 0xFFFFFFu, "embedded masks need fewer than 0xFFFFFF nodes (IntersectorOctreeGPU.hpp:231), got %u", numberOfNodes );
	hipStream_t st = (hipStream_t)stream;
	if( ownerDrain( svo ) ) return 1; // steps already issued keep the octree they were issued with
	svo->cleanUp();
	DevBuf raw;
	if( raw.alloc( (uint64_t)numberOfNodes * 68 ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( raw.p, nodes68Host, (uint64_t)numberOfNodes * 68, hipMemcpyHostToDevice, st ) );
	MVRT_HIP( hipMalloc( (void**)&svo->nodes, (uint64_t)numberOfNodes * sizeof( Node64 ) ) );
	MVRT_HIP( hipMalloc( (void**)&svo->masks, numberOfNodes ) );
	MVRT_HIP( hipMalloc( (void**)&svo->attrs, (uint64_t)( numberOfVoxels ? numberOfVoxels : 1 ) * 8 ) );
	if( numberOfVoxels ) MVRT_HIP( hipMemcpyAsync( svo->attrs, attribs8Host, (uint64_t)numberOfVoxels * 8, hipMemcpyHostToDevice, st ) );
	if( !embeddedMask ) MVRT_HIP( hipMalloc( (void**)&svo->psumCold, (uint64_t)numberOfNodes * 32 ) );
	if( launchConvertNodes( raw.as<uint8_t>(), numberOfNodes, svo->nodes, svo->masks, svo->psumCold, embeddedMask ? 0 : 1, st ) ) return 1;
	svo->info.numberOfNodes = numberOfNodes;
	svo->info.numberOfVoxels = numberOfVoxels;
	svo->info.hasEmission = hasEmission ? 1 : 0;
	svo->info.embeddedMask = embeddedMask ? 1 : 0;
	svo->info.totalDumpedVoxels = 0;
	setBounds( svo, origin, dps, gridRes );
	svo->rootMask = ( (const uint8_t*)nodes68Host )[(uint64_t)( numberOfNodes - 1 ) * 68];
	svo->leafPsumIsPopcount = 1;
	if( embeddedMask ) // an uploaded octree may carry any nVoxelsPSum (mvrt.h): the popcount shortcut of the last level only for canonical ones
	{
		DevBuf bad;
		if( bad.alloc( 4 ) ) return 1;
		if( launchCheckLeafPsum( svo->nodes, svo->masks, numberOfNodes, bad.as<uint32_t>(), st ) ) return 1;
		uint32_t h = 0;
		MVRT_HIP( hipMemcpyAsync( &h, bad.p, 4, hipMemcpyDeviceToHost, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );
		svo->leafPsumIsPopcount = h ? 0u : 1u;
	}
	if( buildTopTable( svo, st ) ) return 1;
	MVRT_HIP( hipStreamSynchronize( st ) );
	return 0;
}

// SvoDev::cellBlocks / cellEntries from the sorted voxel codes of a build (not for the tree flavour, whose traversal reports voxel indices itself)
static int buildCellIndex( mvrt_svo* s, hipStream_t st )
{
	s->cellBlocks.release();
	s->cellEntries.release();
	s->cellBits = 0;
	static const int on = (int)mvrtKnob( "MVRT_CELL_INDEX", 1 );
	const uint32_t L = s->info.levels;
	if( !on || s->tree || !s->morton || s->info.numberOfVoxels == 0 || L == 0 || L > 14u ) return 0;
	static const uint32_t blockBits = (uint32_t)mvrtKnob( "MVRT_CELL_BITS", 9 );
	const uint32_t cellBits = 3u * ( L - 1u ) < blockBits ? 3u * ( L - 1u ) : blockBits; // 8 x 8 x 8 cells per block (fewer in octrees of fewer than 4 levels)
	const uint64_t nBlockCodes = 1ull << ( 3u * ( L - 1u ) - cellBits );
	DevBuf cnt;
	size_t freeB0 = 0, totalB0 = 0;
	if( hipMemGetInfo( &freeB0, &totalB0 ) != hipSuccess || nBlockCodes * 4 > freeB0 / 4 ) return 0;
	if( cnt.alloc( 4 ) || s->cellBlocks.alloc( nBlockCodes * 4 ) ) return 1;
	MVRT_HIP( hipMemsetAsync( cnt.p, 0, 4, st ) );
	MVRT_HIP( hipMemsetAsync( s->cellBlocks.p, 0xFF, nBlockCodes * 4, st ) );
	if( launchNumberCellBlocks( s->morton, s->info.numberOfVoxels, cellBits, s->cellBlocks.as<uint32_t>(), cnt.as<uint32_t>(), st ) ) return 1;
	uint32_t nBlocks = 0;
	MVRT_HIP( hipMemcpyAsync( &nBlocks, cnt.p, 4, hipMemcpyDeviceToHost, st ) );
	MVRT_HIP( hipStreamSynchronize( st ) );
	const uint64_t bytes = ( (uint64_t)nBlocks << cellBits ) * sizeof( uint2 );
	// an accelerator, not a necessity: where it would take more than a quarter of what is free (or cannot be had), the octree keeps the nVoxelsPSum walk
	size_t freeB = 0, totalB = 0;
	if( hipMemGetInfo( &freeB, &totalB ) != hipSuccess || bytes > freeB / 4 || s->cellEntries.alloc( bytes ) )
	{
		(void)hipGetLastError();
		s->cellBlocks.release();
		s->cellEntries.release();
		return 0;
	}
	MVRT_HIP( hipMemsetAsync( s->cellEntries.p, 0, bytes, st ) );
	if( launchFillCellIndex( s->morton, s->info.numberOfVoxels, cellBits, s->cellBlocks.as<uint32_t>(), s->cellEntries.as<uint2>(), st ) ) return 1;
	s->cellBits = cellBits;
	return 0;
}

static int adoptBuild( mvrt_svo* svo, const SvoBuildResult& r, const float origin[3], float dps, int gridRes )
{
	svo->nodes = r.nodes;
	svo->masks = r.masks;
	svo->psumCold = r.psumCold;
	svo->tree = r.tree;
	svo->treeRoot = r.treeRoot;
	svo->nBricks = r.nBricks;
	svo->treeFirst = r.treeFirst;
	memcpy( svo->treeLevelBase, r.treeLevelBase, sizeof( svo->treeLevelBase ) );
	memcpy( svo->treeLevelCount, r.treeLevelCount, sizeof( svo->treeLevelCount ) );
	svo->attrs = r.attrs;
	svo->morton = r.morton;
	svo->info.numberOfNodes = r.nNodes;
	svo->info.numberOfVoxels = r.nVoxels;
	svo->info.hasEmission = r.hasEmission;
	svo->info.embeddedMask = r.embedded; // 0 when the octree has >= 0xFFFFFF nodes (IntersectorOctreeGPU.hpp:231) or on request
	svo->info.totalDumpedVoxels = r.totalDumped;
	svo->leafPsumIsPopcount = 1; // (this library's builder writes the exclusive popcounts)
	setBounds( svo, origin, dps, gridRes );
	MVRT_HIP( hipMemcpy( &svo->rootMask, svo->masks + ( r.nNodes - 1 ), 1, hipMemcpyDeviceToHost ) );
	if( buildTopTable( svo, nullptr ) ) return 1;
	if( buildCellIndex( svo, nullptr ) ) return 1;
	MVRT_HIP( hipDeviceSynchronize() );
	return 0;
}
MVRT_EXPORT int mvrt_svo_build_ex( mvrt_svo* svo, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
								   const float origin[3], float dps, int gridRes, int flags )
{
	REQUIRE( svo && verticesHost && nVertices >= 3 && nVertices % 3 == 0, "mvrt_svo_build: need 3*k vertices" );
	REQUIRE( ilog2Exact( gridRes ) > 0, "gridRes %d is not a power of two >= 2 (IntersectorOctreeGPU.hpp:48-51)", gridRes );
	hipStream_t st = (hipStream_t)stream;
	if( ownerDrain( svo ) ) return 1; // steps already issued keep the octree they were issued with
	svo->cleanUp(); // :53
	SvoBuildResult r;
	memset( &r, 0, sizeof( r ) );
	if( svoBuildFromTriangles( verticesHost, vcolorsHost, vemissionsHost, nVertices, mk3( origin[0], origin[1], origin[2] ), dps, gridRes, flags, st, &r ) ) return 1;
	return adoptBuild( svo, r, origin, dps, gridRes );
}
MVRT_EXPORT int mvrt_svo_build( mvrt_svo* svo, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
								const float origin[3], float dps, int gridRes )
{
	return mvrt_svo_build_ex( svo, verticesHost, vcolorsHost, vemissionsHost, nVertices, stream, origin, dps, gridRes, 0 );
}
MVRT_EXPORT int mvrt_svo_build_synthetic( mvrt_svo* svo, int gridRes, uint64_t nRandomVoxels, uint64_t seed, const float origin[3], float dps, int flags, void* stream )
{
	REQUIRE( svo, "null argument" );
	REQUIRE( ilog2Exact( gridRes ) > 0 && gridRes <= ( 1 << 21 ), "gridRes %d is not a power of two in [2, 2^21]", gridRes );
	if( ownerDrain( svo ) ) return 1;
	svo->cleanUp();
	SvoBuildResult r;
	memset( &r, 0, sizeof( r ) );
	if( svoBuildSynthetic( nRandomVoxels, seed, gridRes, flags, (hipStream_t)stream, &r ) ) return 1;
	return adoptBuild( svo, r, origin, dps, gridRes );
}

MVRT_EXPORT int mvrt_svo_get_info( const mvrt_svo* svo, mvrt_svo_info* info )
{
	REQUIRE( svo && info, "null argument" );
	*info = svo->info;
	info->flavour = svo->tree ? MVRT_FLAVOUR_TREE : ( svo->info.embeddedMask ? MVRT_FLAVOUR_EMBEDDED : MVRT_FLAVOUR_PLAIN );
	info->reserved = 0;
	return 0;
}
MVRT_EXPORT uint64_t mvrt_svo_traversal_bytes( const mvrt_svo* svo )
{
	if( !svo || !svo->nodes ) return 0;
	const uint64_t n = svo->info.numberOfNodes;
	if( svo->tree ) return (uint64_t)svo->nBricks * sizeof( uint4 ) + n * 5;
	return n * sizeof( Node64 ) + n + ( svo->psumCold ? n * 32 : 0 ) + svo->topTable.bytes + svo->kids.bytes + svo->cellBlocks.bytes + svo->cellEntries.bytes;
}
MVRT_EXPORT const void* mvrt_svo_node_buffer_dev( const mvrt_svo* svo ) { return svo ? svo->nodes : nullptr; }
MVRT_EXPORT const void* mvrt_svo_attribute_buffer_dev( const mvrt_svo* svo ) { return svo ? svo->attrs : nullptr; }
MVRT_EXPORT int mvrt_svo_set_emission_scale( mvrt_svo* svo, float scale )
{
	REQUIRE( svo, "null argument" );
	if( ownerFlush( svo ) ) return 1; // pending steps are launched with the scale they were issued under
	svo->info.emissionScale = scale;
	return 0;
}
MVRT_EXPORT int mvrt_svo_download( const mvrt_svo* svo, void* nodes68Host, void* attribs8Host, uint64_t* mortonHost, void* stream )
{
	REQUIRE( svo && svo->nodes, "no octree" );
	hipStream_t st = (hipStream_t)stream;
	if( nodes68Host )
	{
		DevBuf raw;
		if( raw.alloc( (uint64_t)svo->info.numberOfNodes * 68 ) ) return 1;
		if( svo->tree )
		{
			if( launchTreeTo68( svo->masks, svo->treeFirst, svo->treeLevelBase, svo->treeLevelCount, (int)svo->info.levels, svo->info.numberOfNodes, svo->info.numberOfVoxels,
								raw.as<uint8_t>(), st ) )
				return 1;
		}
		else if( launchNodesTo68( svo->nodes, svo->masks, svo->psumCold, svo->info.numberOfNodes, raw.as<uint8_t>(), svo->info.embeddedMask ? 0 : 1, st ) ) return 1;
		MVRT_HIP( hipMemcpyAsync( nodes68Host, raw.p, raw.bytes, hipMemcpyDeviceToHost, st ) );
		MVRT_HIP( hipStreamSynchronize( st ) );
	}
	if( attribs8Host ) MVRT_HIP( hipMemcpyAsync( attribs8Host, svo->attrs, (uint64_t)svo->info.numberOfVoxels * 8, hipMemcpyDeviceToHost, st ) );
	if( mortonHost )
	{
		REQUIRE( svo->morton, "morton codes are only kept by mvrt_svo_build" );
		MVRT_HIP( hipMemcpyAsync( mortonHost, svo->morton, (uint64_t)svo->info.numberOfVoxels * 8, hipMemcpyDeviceToHost, st ) );
	}
	MVRT_HIP( hipStreamSynchronize( st ) );
	return 0;
}

MVRT_EXPORT int mvrt_trace_batch( const mvrt_svo* svo, uint64_t n, const float* roxDev, const float* royDev, const float* rozDev, const float* rdxDev, const float* rdyDev,
								  const float* rdzDev, const uint8_t* isShadowDev, float* tDev, int32_t* nMajorDev, uint32_t* vIndexDev, uint32_t* descentsDev, void* stream )
{
	REQUIRE( svo && svo->nodes, "mvrt_trace_batch: no octree (build or upload first)" );
	REQUIRE( tDev, "mvrt_trace_batch: t output is required" );
	if( svo->ensureWorkspace( vIndexDev ? n : 0 ) ) return 1;
	return launchTraceBatch( svo->dev(), svo->ws, n, roxDev, royDev, rozDev, rdxDev, rdyDev, rdzDev, isShadowDev, tDev, nMajorDev, vIndexDev, descentsDev, (hipStream_t)stream );
}

MVRT_EXPORT int mvrt_trace_batch_hinted( const mvrt_svo* svo, uint64_t n, const float* roxDev, const float* royDev, const float* rozDev, const float* rdxDev, const float* rdyDev,
										 const float* rdzDev, const uint8_t* isShadowDev, const uint64_t* originVoxelMortonDev, float* tDev, int32_t* nMajorDev, uint32_t* vIndexDev,
										 uint32_t* descentsDev, void* stream )
{
	REQUIRE( svo && svo->nodes, "mvrt_trace_batch_hinted: no octree (build or upload first)" );
	REQUIRE( tDev, "mvrt_trace_batch_hinted: t output is required" );
	if( svo->ensureWorkspace( vIndexDev ? n : 0 ) ) return 1;
	return launchTraceBatch( svo->dev(), svo->ws, n, roxDev, royDev, rozDev, rdxDev, rdyDev, rdzDev, isShadowDev, tDev, nMajorDev, vIndexDev, descentsDev, (hipStream_t)stream,
							 originVoxelMortonDev );
}

MVRT_EXPORT int mvrt_trace_batch_host( const mvrt_svo* svo, uint64_t n, const float* roHost, const float* rdHost, const uint8_t* isShadowHost, float* tHost,
									   int32_t* nMajorHost, uint32_t* vIndexHost, uint32_t* descentsHost )
{
	REQUIRE( svo && svo->nodes, "mvrt_trace_batch_host: no octree" );
	if( n == 0 ) return 0;
	std::vector<float> soa( n * 6 );
	for( uint64_t i = 0; i < n; i++ )
		for( int k = 0; k < 3; k++ )
		{
			soa[k * n + i] = roHost[i * 3 + k];
			soa[( 3 + k ) * n + i] = rdHost[i * 3 + k];
		}
	DevBuf in, sh, t, nm, vi, de;
	if( in.alloc( n * 24 ) || t.alloc( n * 4 ) || nm.alloc( n * 4 ) || vi.alloc( n * 4 ) || de.alloc( n * 4 ) ) return 1;
	MVRT_HIP( hipMemcpy( in.p, soa.data(), n * 24, hipMemcpyHostToDevice ) );
	if( isShadowHost )
	{
		if( sh.alloc( n ) ) return 1;
		MVRT_HIP( hipMemcpy( sh.p, isShadowHost, n, hipMemcpyHostToDevice ) );
	}
	const float* b = in.as<float>();
	if( svo->ensureWorkspace( n ) ) return 1;
	if( launchTraceBatch( svo->dev(), svo->ws, n, b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 5 * n, isShadowHost ? sh.as<uint8_t>() : nullptr, t.as<float>(), nm.as<int32_t>(),
						  vi.as<uint32_t>(), de.as<uint32_t>(), 0 ) )
		return 1;
	MVRT_HIP( hipDeviceSynchronize() );
	MVRT_HIP( hipMemcpy( tHost, t.p, n * 4, hipMemcpyDeviceToHost ) );
	if( nMajorHost ) MVRT_HIP( hipMemcpy( nMajorHost, nm.p, n * 4, hipMemcpyDeviceToHost ) );
	if( vIndexHost ) MVRT_HIP( hipMemcpy( vIndexHost, vi.p, n * 4, hipMemcpyDeviceToHost ) );
	if( descentsHost ) MVRT_HIP( hipMemcpy( descentsHost, de.p, n * 4, hipMemcpyDeviceToHost ) );
	return 0;
}

static CameraPinhole cameraFrom15( const float c[15] )
{
	CameraPinhole cam;
	memcpy( &cam, c, sizeof( cam ) );
	return cam;
}

MVRT_EXPORT int mvrt_render_primary( const mvrt_svo* svo, const float camera[15], int width, int height, int showVertexColor, uint8_t* rgbaDev, float* tDev,
									 int32_t* nMajorDev, uint32_t* vIndexDev, uint32_t* descentsDev, void* stream )
{
	REQUIRE( svo && svo->nodes, "mvrt_render_primary: no octree" );
	REQUIRE( width > 0 && height > 0, "bad resolution %dx%d", width, height );
	if( svo->ensureWorkspace( ( vIndexDev || showVertexColor ) ? (uint64_t)width * height : 0 ) ) return 1;
	if( !tDev && ( vIndexDev || showVertexColor ) ) tDev = (float*)( svo->ws.paths + svo->ws.pathCap ); // the resolve pass needs t
	return launchRenderPrimary( svo->dev(), svo->ws, cameraFrom15( camera ), width, height, showVertexColor, (uchar4*)rgbaDev, tDev, nMajorDev, vIndexDev, descentsDev,
								(hipStream_t)stream );
}

// CameraPinhole::initFromPerspective, renderCommon.hpp:21-35 (glm column-major matrices)
MVRT_EXPORT int mvrt_camera_from_matrices( const float view[16], const float proj[16], float focus, float lensR, float cameraOut[15] )
{
	f3 r0 = mk3( view[0], view[4], view[8] ); // rows of mat3(view) = columns of its transpose
	f3 r1 = mk3( view[1], view[5], view[9] );
	f3 r2 = mk3( view[2], view[6], view[10] );
	f3 v = mk3( view[12], view[13], view[14] );
	f3 m = mk3( r0.x * v.x + r1.x * v.y + r2.x * v.z, r0.y * v.x + r1.y * v.y + r2.y * v.z, r0.z * v.x + r1.z * v.y + r2.z * v.z );
	CameraPinhole c;
	c.front = mk3( -r2.x, -r2.y, -r2.z );
	c.up = r1;
	c.right = r0;
	c.o = mk3( -m.x, -m.y, -m.z );
	c.tanHthetaY = 1.0f / proj[5];
	c.lensR = lensR;
	c.focus = focus;
	memcpy( cameraOut, &c, sizeof( c ) );
	return 0;
}

MVRT_EXPORT int mvrt_compact_indices( const uint8_t* keepDev, uint64_t n, uint32_t* dstIndexDev, uint32_t* keptDev, void* stream )
{
	if( n == 0 )
	{
		if( keptDev ) MVRT_HIP( hipMemsetAsync( keptDev, 0, 4, (hipStream_t)stream ) );
		return 0;
	}
	DevBuf scratch;
	if( scratch.alloc( ( n / 256 + 8 ) * 4 ) ) return 1; // (+ padding: the scan reads whole 16-byte quads)
	if( launchCompactIndices( keepDev, n, dstIndexDev, keptDev, scratch.as<uint32_t>(), (hipStream_t)stream ) ) return 1;
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) ); // scratch is freed on return
	return 0;
}

// ---- PMJ02 table (host), pmjSampler.hpp:14-58,114-144 -----------------------------------------------------
namespace
{
struct PCG32 // renderCommon.hpp:86-110
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
		uint64_t old = state;
		state = old * 6364136223846793005ULL + inc;
		uint32_t xs = (uint32_t)( ( ( old >> 18u ) ^ old ) >> 27u );
		uint32_t rot = (uint32_t)( old >> 59u );
		return ( xs >> rot ) | ( xs << ( ( -rot ) & 31 ) );
	}
};
inline float uniformf( uint32_t x ) { return mvrt_u2f( ( x >> 9 ) | 0x3f800000u ) - 1.0f; } // :112-117

// Helmer's stochastic generation of a pmj02 sequence; xi0 is drawn before xi1 (see DESIGN.md "PMJ order")
void pmj02Sequence( int numSamples, float* samples, PCG32& rng )
{
	static const uint32_t xors[2][32] = {
		{ 0x0, 0x0, 0x2, 0x6, 0x6, 0xe, 0x36, 0x4e, 0x16, 0x2e, 0x276, 0x6ce, 0x716, 0xc2e, 0x3076, 0x40ce, 0x116, 0x22e, 0x20676, 0x60ece, 0x61716,
		  0xe2c2e, 0x367076, 0x4ec0ce, 0x170116, 0x2c022e, 0x2700676, 0x6c00ece, 0x7001716, 0xc002c2e, 0x30007076, 0x4000c0ce },
		{ 0x0, 0x1, 0x3, 0x3, 0x7, 0x1b, 0x27, 0xb, 0x17, 0x13b, 0x367, 0x38b, 0x617, 0x183b, 0x2067, 0x8b, 0x117, 0x1033b, 0x30767, 0x30b8b,
		  0x71617, 0x1b383b, 0x276067, 0xb808b, 0x160117, 0x138033b, 0x3600767, 0x3800b8b, 0x6001617, 0x1800383b, 0x20006067, 0x808b } };
	samples[0] = uniformf( rng.nextU32() );
	samples[1] = uniformf( rng.nextU32() );
	for( int logN = 0; ( 1 << logN ) < numSamples; logN++ )
	{
		const int prevLen = 1 << logN;
		const int nStrata = prevLen * 2;
		const float iStrata = 1.0f / nStrata;
		for( int i = 0; i < prevLen && ( prevLen + i ) < numSamples; i++ )
		{
			const int xStratum = ( (int)( samples[( i ^ xors[0][logN] ) * 2] * nStrata ) ) ^ 1;
			const int yStratum = ( (int)( samples[( i ^ xors[1][logN] ) * 2 + 1] * nStrata ) ) ^ 1;
			const float xi0 = uniformf( rng.nextU32() );
			const float xi1 = uniformf( rng.nextU32() );
			samples[( prevLen + i ) * 2] = ( xi0 + xStratum ) * iStrata;
			samples[( prevLen + i ) * 2 + 1] = ( xi1 + yStratum ) * iStrata;
		}
	}
}

// Radiance RGBE reader: header lines, blank line, "-Y h +X w", then flat or new-RLE scanlines.
int loadRgbe( const char* path, std::vector<float>& rgba, int* w, int* h )
{
	FILE* fp = fopen( path, "rb" );
	REQUIRE( fp, "cannot open %s", path );
	std::vector<uint8_t> d;
	uint8_t tmp[65536];
	size_t got;
	while( ( got = fread( tmp, 1, sizeof( tmp ), fp ) ) > 0 ) d.insert( d.end(), tmp, tmp + got );
	fclose( fp );
	size_t pos = 0;
	bool blank = false;
	*w = *h = 0;
	while( pos < d.size() )
	{
		size_t e = pos;
		while( e < d.size() && d[e] != '\n' ) e++;
		std::string line( (const char*)d.data() + pos, e - pos );
		pos = e + 1;
		if( !blank )
		{
			if( line.empty() ) blank = true;
			continue;
		}
		REQUIRE( sscanf( line.c_str(), "-Y %d +X %d", h, w ) == 2, "%s: unsupported resolution line '%s'", path, line.c_str() );
		break;
	}
	REQUIRE( *w > 0 && *h > 0, "%s: not a Radiance .hdr file", path );
	const int W = *w, H = *h;
	rgba.resize( (size_t)W * H * 4 );
	std::vector<uint8_t> scan( (size_t)W * 4 );
	for( int y = 0; y < H; y++ )
	{
		if( pos + 4 <= d.size() && W >= 8 && W < 32768 && d[pos] == 2 && d[pos + 1] == 2 && ( d[pos + 2] & 0x80 ) == 0 && ( ( d[pos + 2] << 8 ) | d[pos + 3] ) == W )
		{
			pos += 4;
			for( int c = 0; c < 4; c++ )
			{
				int x = 0;
				while( x < W )
				{
					REQUIRE( pos < d.size(), "%s: truncated RLE data", path );
					int count = d[pos++];
					if( count > 128 )
					{
						count -= 128;
						REQUIRE( pos < d.size() && x + count <= W, "%s: bad RLE run", path );
						uint8_t v = d[pos++];
						for( int k = 0; k < count; k++ ) scan[( x++ ) * 4 + c] = v;
					}
					else
					{
						REQUIRE( pos + count <= d.size() && x + count <= W, "%s: bad RLE literal", path );
						for( int k = 0; k < count; k++ ) scan[( x++ ) * 4 + c] = d[pos++];
					}
				}
			}
		}
		else
		{
			REQUIRE( pos + (size_t)W * 4 <= d.size(), "%s: truncated pixel data", path );
			memcpy( scan.data(), d.data() + pos, (size_t)W * 4 );
			pos += (size_t)W * 4;
		}
		for( int x = 0; x < W; x++ )
		{
			const uint8_t* p = &scan[(size_t)x * 4];
			float* o = &rgba[( (size_t)y * W + x ) * 4];
			if( p[3] )
			{
				float f = mvrt_u2f( (uint32_t)( (int)p[3] - 136 + 127 ) << 23 ); // 2^(E-136), E >= 10 keeps it normal
				if( (int)p[3] - 136 + 127 <= 0 ) f = 0.0f;
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
} // namespace

// ---- PathTracer ---------------------------------------------------------------------------------------------
struct EventProfiler : PtProfiler
{
	struct Rec
	{
		hipEvent_t a, b;
		int cls;
	};
	std::vector<Rec> recs;
	std::vector<hipEvent_t> pool;
	double ms[3] = { 0, 0, 0 };
	uint64_t traceLaunches = 0;
	hipEvent_t get()
	{
		if( !pool.empty() )
		{
			hipEvent_t e = pool.back();
			pool.pop_back();
			return e;
		}
		hipEvent_t e;
		(void)hipEventCreate( &e );
		return e;
	}
	void begin( int cls, hipStream_t s ) override
	{
		Rec r;
		r.a = get();
		r.b = get();
		r.cls = cls;
		(void)hipEventRecord( r.a, s );
		recs.push_back( r );
	}
	void end( hipStream_t s ) override { (void)hipEventRecord( recs.back().b, s ); }
	void collect() // caller has synchronised the stream
	{
		for( Rec& r : recs )
		{
			float t = 0.0f;
			if( hipEventElapsedTime( &t, r.a, r.b ) == hipSuccess ) ms[r.cls] += t;
			if( r.cls == MVRT_K_TRACE ) traceLaunches++;
			pool.push_back( r.a );
			pool.push_back( r.b );
		}
		recs.clear();
	}
	~EventProfiler()
	{
		collect();
		for( hipEvent_t e : pool ) (void)hipEventDestroy( e );
	}
};

struct mvrt_pt
{
	mvrt_svo* intersector = nullptr; // m_intersectorOctreeGPU
	DevBuf pmj;						 // m_pmj
	// m_hdri
	DevBuf hdriPixels, hdriPrimary, hdriSat[7];
	HdriDev hdri;
	// frame buffers
	DevBuf fbF32, fbU8; // m_frameBufferF32 / m_frameBufferU8
	int width = 0, height = 0, steps = 0;
	int tileIndex = 0, tileCount = 1;
	uint64_t ownedPixels = 0, validOwnedPixels = 0;
	// wavefront work buffers: one set per in-flight step.  Consecutive step() calls are pipelined on internal
	// streams (depth slots) so that the thin late bounces of one step overlap the dense early bounces of the next;
	// the frame-buffer additions stay in step order through an event chain.  depth 1 = everything on the caller's stream.
	struct Slot
	{
		DevBuf work, wsBuf, dbg;
		PtBuffers buf;
		TraceWorkspace ws = { nullptr, 0, nullptr, nullptr, 0, nullptr };
		hipStream_t stream = nullptr;
		hipEvent_t accumDone = nullptr;
	};
	Slot slots[4];
	// deferred execution: up to `batch` consecutive step() calls are merged into ONE wavefront pass (bigger launches,
	// same per-sample results, frame-buffer additions still in step order).  Flushed by any consumer of the frame.
	std::vector<CameraPinhole> pendingCams;
	int pendingIteration = 0;
	hipStream_t pendingStream = nullptr;
	int batchCap = MVRT_MAX_BATCH; // footprint bound (allocWork)
	int batch = 0; // 0 = automatic: merge steps until a pass holds ~2 full-HD steps worth of samples (see effectiveBatch)
	int depth = 3;
	int depthWanted = 3; // what the caller asked for; `depth` may be lower when the path state would not fit (allocWork)
	int nextSlot = 0, lastSlot = 0;
	DevBuf statsBuf;
	hipEvent_t forkEv = nullptr;
	hipEvent_t lastAccum = nullptr;
	bool pendingJoin = false;
	PtBuffers& buf = slots[0].buf; // slot 0 doubles as "the" buffer set for capacity / stats bookkeeping
	bool setupDone = false;
	bool profiling = false;
	uint64_t testFreeBytes = 0; // != 0: allocWork budgets against this instead of hipMemGetInfo (tests of the failure path)
	bool debugCapture = false; // keep the survivor list of every shade stage of the last pass (mvrt_pt_set_debug_capture)
	EventProfiler prof;
	int numCUs = 0;
	mvrt_pt()
	{
		memset( &hdri, 0, sizeof( hdri ) );
		hdri.scale = 1.75f; // renderCommon.hpp:480
		for( Slot& sl : slots ) memset( &sl.buf, 0, sizeof( sl.buf ) );
		depth = (int)mvrtKnob( "MVRT_PIPELINE_DEPTH", depth );
		if( depth < 1 ) depth = 1;
		if( depth > 4 ) depth = 4;
		depthWanted = depth;
		splitSmallPasses = mvrtKnob( "MVRT_SPLIT_SMALL", 1 ) != 0;
		batch = (int)mvrtKnob( "MVRT_BATCH_STEPS", 0 );
		if( batch < 0 ) batch = 0;
		if( batch > MVRT_MAX_BATCH ) batch = MVRT_MAX_BATCH;
	}
	int flush( bool moreStepsFollow = false ); // launch the pending steps (defined below)
	int launchPass( const CameraPinhole* passCams, int iteration, int nSteps, int traceGridDiv );
	bool splitSmallPasses = true; // MVRT_SPLIT_SMALL=0 disables (A/B)
	bool originHints = true;	  // secondary rays start below the root (mvrt_pt_set_origin_hints)
	int effectiveBatch() const // merged steps per pass, bounded so that one pass stays below ~160 M samples (~30 GB of path state)
	{
		uint64_t perStep = ownedPixels * MVRT_SPP_PER_STEP;
		if( perStep == 0 ) return 1;
		int b = batch;
		if( b == 0 ) // automatic: big enough to amortise launch tails, small enough that several passes can pipeline
		{
			b = (int)( ( 66000000ull + perStep / 2 ) / perStep );
			if( b < 1 ) b = 1;
			if( b > MVRT_MAX_BATCH ) b = MVRT_MAX_BATCH;
		}
		while( b > 1 && perStep * b > 160000000ull ) b--;
		if( b > batchCap ) b = batchCap; // lowered by allocWork when the path state would not fit the free HBM
		return b;
	}
	// steps merged into the next pass: effectiveBatch(), but at most HALF of the caller's frame -- the number of steps it accumulated before its last
	// clearFrameBuffer -- when no batch size was set: a frame that fits ONE pass has nothing to overlap that pass's launch tails and shade kernels with, two passes
	// hide each other's (measured on a 1/2 tile share of a 64-spp frame, where the automatic batch is the whole frame: 9.23 -> 8.49 ms per step; full frame and
	// 1/4, 1/8 shares unchanged).  Capacity is sized by effectiveBatch(), which this never exceeds.
	int lastFrameSteps = 0;
	int passSteps() const
	{
		int b = effectiveBatch();
		if( batch == 0 && lastFrameSteps >= 2 && b > ( lastFrameSteps + 1 ) / 2 ) b = ( lastFrameSteps + 1 ) / 2;
		return b;
	}
	// make `user` stream wait for every step that was issued on the internal streams
	int join( hipStream_t user )
	{
		if( flush() ) return 1;
		if( pendingJoin && lastAccum ) MVRT_HIP( hipStreamWaitEvent( user, lastAccum, 0 ) );
		pendingJoin = false;
		return 0;
	}
	int drain() // host-side: everything the internal streams hold has finished
	{
		if( flush() ) return 1;
		// depth 1 runs on the caller's stream, which the caller may have destroyed since: wait for the event recorded behind the pass
		if( lastAccum ) MVRT_HIP( hipEventSynchronize( lastAccum ) );
		for( Slot& sl : slots )
			if( sl.stream ) MVRT_HIP( hipStreamSynchronize( sl.stream ) );
		pendingJoin = false;
		pendingStream = nullptr;
		return 0;
	}
	~mvrt_pt()
	{
		for( Slot& sl : slots )
		{
			if( sl.stream )
			{
				(void)hipStreamSynchronize( sl.stream );
				(void)hipStreamDestroy( sl.stream );
			}
			if( sl.accumDone ) (void)hipEventDestroy( sl.accumDone );
		}
		if( forkEv ) (void)hipEventDestroy( forkEv );
	}
};

static int ptFlush( mvrt_pt* pt ) { return pt->flush(); }
static int ptDrain( mvrt_pt* pt ) { return pt->drain(); }

MVRT_EXPORT int mvrt_pt_create( mvrt_pt** out )
{
	mvrt_pt* pt = new mvrt_pt();
	pt->intersector = new mvrt_svo();
	pt->intersector->owner = pt;
	*out = pt;
	return 0;
}
MVRT_EXPORT int mvrt_pt_destroy( mvrt_pt* pt )
{
	if( pt )
	{
		mvrt_svo_destroy( pt->intersector );
		delete pt;
	}
	return 0;
}
MVRT_EXPORT int mvrt_pt_setup( mvrt_pt* pt, void* stream )
{
	REQUIRE( pt, "null argument" );
	// PMJSampler::setup, pmjSampler.hpp:114-144
	std::vector<float> samples( (size_t)2 * MVRT_PMJ_LENGTH * MVRT_PMJ_NSEQ );
	PCG32 rng;
	rng.setup( 0, 2525 );
	for( int i = 0; i < MVRT_PMJ_NSEQ; i++ ) pmj02Sequence( MVRT_PMJ_LENGTH, samples.data() + (size_t)2 * MVRT_PMJ_LENGTH * i, rng );
	if( pt->pmj.alloc( samples.size() * 4 ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( pt->pmj.p, samples.data(), samples.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream ) );
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	int dev = 0;
	MVRT_HIP( hipGetDevice( &dev ) );
	hipDeviceProp_t p;
	MVRT_HIP( hipGetDeviceProperties( &p, dev ) );
	pt->numCUs = p.multiProcessorCount;
	pt->setupDone = true;
	return 0;
}

MVRT_EXPORT int mvrt_pt_download_pmj( mvrt_pt* pt, float* tableHost )
{
	REQUIRE( pt && pt->pmj.p && tableHost, "mvrt_pt_download_pmj: call mvrt_pt_setup first" );
	MVRT_HIP( hipMemcpy( tableHost, pt->pmj.p, pt->pmj.bytes, hipMemcpyDeviceToHost ) );
	return 0;
}

MVRT_EXPORT int mvrt_pt_set_tile( mvrt_pt* pt, int tileIndex, int tileCount )
{
	REQUIRE( pt && tileCount >= 1 && tileIndex >= 0 && tileIndex < tileCount, "bad tile %d of %d", tileIndex, tileCount );
	if( pt->drain() ) return 1;
	pt->tileIndex = tileIndex;
	pt->tileCount = tileCount;
	pt->width = pt->height = 0; // force re-allocation on the next resize
	pt->fbF32.release();
	return 0;
}
MVRT_EXPORT uint64_t mvrt_pt_owned_pixels( const mvrt_pt* pt ) { return pt ? pt->ownedPixels : 0; }

static int allocWorkSlot( mvrt_pt* pt, mvrt_pt::Slot& sl )
{
	const uint64_t cap = pt->ownedPixels * MVRT_SPP_PER_STEP * pt->effectiveBatch();
	const uint64_t nBlocks = cap / 256 + 8; // (+ padding: the scan reads whole 16-byte quads)
	// 2 path sets x 17 arrays + 6 direction arrays + hitT + 3 sample-radiance arrays + 2 x 8-byte path arrays,
	// 4 bytes per word; 3 byte arrays; every array is padded to 256 B so the float4 reads of Ls* stay aligned
	const uint64_t words = cap * ( 34 + 6 + 1 + 3 + 4 );
	const uint64_t bytes = words * 4 + cap * 3 + nBlocks * 4 + 64 * 4 + 32 * 8 + 1024 + 64 * 256;
	if( sl.work.alloc( bytes ) ) return 1;
	uint8_t* base = (uint8_t*)sl.work.p;
	uint64_t off = 0;
	auto take = [&]( uint64_t b ) {
		void* r = base + off;
		off += ( b + 255 ) & ~(uint64_t)255;
		return r;
	};
	PtBuffers& b = sl.buf;
	for( int s = 0; s < 2; s++ )
	{
		PathSet& ps = b.set[s];
		ps.task = (uint32_t*)take( cap * 4 );
		ps.org = (uint32_t*)take( cap * 4 );
		float** f[] = { &ps.rox, &ps.roy, &ps.roz, &ps.rdx, &ps.rdy, &ps.rdz, &ps.Tx, &ps.Ty, &ps.Tz, &ps.Lx, &ps.Ly, &ps.Lz, &ps.nx, &ps.ny, &ps.nz };
		for( float** q : f ) *q = (float*)take( cap * 4 );
	}
	float** d[] = { &b.sx, &b.sy, &b.sz, &b.ex, &b.ey, &b.ez, &b.hitT, &b.Lsx, &b.Lsy, &b.Lsz };
	for( float** q : d ) *q = (float*)take( cap * 4 );
	b.hitPath = (uint64_t*)take( cap * 8 );
	b.hitEPath = (uint64_t*)take( cap * 8 );
	b.hitN = (uint8_t*)take( cap );
	b.hitS = (uint8_t*)take( cap );
	b.hitE = (uint8_t*)take( cap );
	b.blockCount = (uint32_t*)take( nBlocks * 4 );
	b.liveCount = (uint32_t*)take( 64 * 4 );
	b.cursors = (unsigned long long*)take( 16 * 8 );
	b.stats = pt->statsBuf.as<unsigned long long>(); // shared by all slots (atomic tallies)
	b.cap = cap;
	b.selfDev = (const PtBuffers*)take( sizeof( PtBuffers ) );
	MVRT_HIP( hipMemset( b.liveCount, 0, 64 * 4 ) );
	MVRT_HIP( hipMemcpy( (void*)b.selfDev, &b, sizeof( PtBuffers ), hipMemcpyHostToDevice ) );
	return 0;
}
static int allocWorkInner( mvrt_pt* pt );
// (Re)allocates the path state of every pipeline slot.  On ANY failure nothing stale is left behind: every slot's buffer table is zeroed (cap = 0, so
// launchPass refuses to run) and the frame is forgotten (width = height = 0, frame buffer released), so that the next
// resizeFrameBufferIfNeeded -- same size or not -- allocates again instead of returning early onto freed memory.
static int allocWork( mvrt_pt* pt )
{
	if( allocWorkInner( pt ) == 0 ) return 0;
	for( mvrt_pt::Slot& sl : pt->slots )
	{
		sl.work.release();
		memset( &sl.buf, 0, sizeof( sl.buf ) );
	}
	pt->width = pt->height = 0;
	pt->fbF32.release();
	pt->fbU8.release();
	return 1;
}
static int allocWorkInner( mvrt_pt* pt )
{
	if( pt->drain() ) return 1;
	if( !pt->statsBuf.p )
	{
		if( pt->statsBuf.alloc( 64 * 8 ) ) return 1;
		MVRT_HIP( hipMemset( pt->statsBuf.p, 0, 64 * 8 ) );
	}
	if( !pt->forkEv ) MVRT_HIP( hipEventCreateWithFlags( &pt->forkEv, hipEventDisableTiming ) );
	// Footprint: every in-flight pass owns ~190 bytes of path state per sample (two ping-pong path sets, ray directions, hit records, per-sample
	// radiance) -- depth x batch x ownedPixels x 16 x 190 B, e.g. 3 x 2 x 6.3 GB at 1920x1080 (the reference: a fixed 1.24 GB stack slab).  It
	// must fit beside the octree: when it would take more than 70 % of the HBM that is free right now, merge fewer steps per pass first, then
	// keep fewer passes in flight.  Results do not depend on either.
	{
		for( mvrt_pt::Slot& sl : pt->slots ) // (what is allocated now is about to be replaced)
		{
			sl.work.release();
			memset( &sl.buf, 0, sizeof( sl.buf ) ); // no pointer into the freed block survives a failure below
		}
		size_t freeB = 0, totalB = 0;
		MVRT_HIP( hipMemGetInfo( &freeB, &totalB ) );
		if( pt->testFreeBytes ) freeB = (size_t)pt->testFreeBytes; // mvrt_pt_set_test_free_bytes: pretend this much HBM is free (failure-path tests)
		pt->batchCap = MVRT_MAX_BATCH;
		pt->depth = pt->depthWanted;
		const uint64_t budget = (uint64_t)( 0.7 * (double)freeB );
		auto need = [&]() { return (uint64_t)pt->depth * pt->ownedPixels * MVRT_SPP_PER_STEP * (uint64_t)pt->effectiveBatch() * 200ull; };
		while( need() > budget && pt->effectiveBatch() > 1 ) pt->batchCap = pt->effectiveBatch() - 1;
		while( need() > budget && pt->depth > 1 ) pt->depth--;
		REQUIRE( need() <= budget, "frame of %llu owned pixels needs %.1f GB of path state, %.1f GB of HBM are free (split the frame into tiles: mvrt_pt_set_tile)",
				 (unsigned long long)pt->ownedPixels, need() / 1e9, freeB / 1e9 );
	}
	for( int i = 0; i < 4; i++ )
	{
		mvrt_pt::Slot& sl = pt->slots[i];
		if( i >= pt->depth )
		{
			sl.work.release();
			sl.wsBuf.release();
			sl.ws = TraceWorkspace{ nullptr, 0, nullptr, nullptr, 0, nullptr };
			memset( &sl.buf, 0, sizeof( sl.buf ) );
			continue;
		}
		if( allocWorkSlot( pt, sl ) ) return 1;
		if( pt->depth > 1 && !sl.stream )
		{
			// (experiment knob: slot streams of different priorities live in different hardware-queue pools and are dispatched in priority order)
			const int mode = (int)mvrtKnob( "MVRT_SLOT_PRIO", 0 );
			int lo = 0, hi = 0;
			MVRT_HIP( hipDeviceGetStreamPriorityRange( &lo, &hi ) );
			const int prio = mode == 0 ? 0 : ( mode == 1 ? ( i == 0 ? hi : ( i == 2 ? lo : 0 ) ) : ( i % 2 == 0 ? hi : lo ) );
			MVRT_HIP( hipStreamCreateWithPriority( &sl.stream, hipStreamNonBlocking, prio ) );
		}
		if( !sl.accumDone ) MVRT_HIP( hipEventCreateWithFlags( &sl.accumDone, hipEventDisableTiming ) );
	}
	pt->nextSlot = pt->lastSlot = 0;
	pt->lastAccum = nullptr;
	return 0;
}
static int ensureSlotWorkspace( mvrt_pt* pt, mvrt_pt::Slot& sl )
{
	const uint64_t lanes = traceWorkspaceLanes();
	const uint64_t rows = 2 * (uint64_t)( pt->intersector->info.levels ? pt->intersector->info.levels : 1 ) + 2;
	const uint64_t bytes = 256 + rows * lanes * ( sizeof( uint4 ) + 2 * sizeof( uint32_t ) );
	if( sl.wsBuf.bytes < bytes )
	{
		if( sl.wsBuf.alloc( bytes ) ) return 1;
	}
	sl.ws.cursor = (unsigned long long*)sl.wsBuf.p;
	sl.ws.spill = (uint4*)( (uint8_t*)sl.wsBuf.p + 256 );
	sl.ws.spillStride = lanes;
	sl.ws.spillMask = (uint32_t*)( sl.ws.spill + rows * lanes );
	sl.ws.spillMask2 = sl.ws.spillMask + rows * lanes;
	return 0;
}

MVRT_EXPORT int mvrt_pt_clear_framebuffer( mvrt_pt* pt, void* stream )
{
	REQUIRE( pt && pt->fbF32.p, "no frame buffer" );
	if( pt->join( (hipStream_t)stream ) ) return 1;
	if( pt->steps >= 2 ) pt->lastFrameSteps = pt->steps; // the caller's frame length (passSteps); a one-step frame says nothing about the next one
	pt->steps = 0; // PathTracer.hpp:100
	MVRT_HIP( hipMemsetAsync( pt->fbF32.p, 0, pt->fbF32.bytes, (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_pt_resize_framebuffer_if_needed( mvrt_pt* pt, void* stream, int width, int height )
{
	REQUIRE( pt && width > 0 && height > 0, "bad resolution %dx%d", width, height );
	if( pt->fbF32.p && pt->width == width && pt->height == height ) return 0;
	if( pt->drain() ) return 1;
	const uint64_t nPix = (uint64_t)width * height;
	const uint64_t nBlocks = ( nPix + MVRT_TILE_PIXELS - 1 ) / MVRT_TILE_PIXELS;
	// blocks b with b % tileCount == tileIndex
	const uint64_t myBlocks = nBlocks > (uint64_t)pt->tileIndex ? ( nBlocks - pt->tileIndex + pt->tileCount - 1 ) / pt->tileCount : 0;
	// every rank pads to the same count so an all-gather of equal chunks works
	const uint64_t maxBlocks = ( nBlocks + pt->tileCount - 1 ) / pt->tileCount;
	pt->ownedPixels = maxBlocks * MVRT_TILE_PIXELS;
	uint64_t valid = myBlocks * MVRT_TILE_PIXELS;
	if( myBlocks > 0 )
	{
		const uint64_t lastGlobalBlock = ( myBlocks - 1 ) * pt->tileCount + pt->tileIndex;
		const uint64_t endPix = ( lastGlobalBlock + 1 ) * MVRT_TILE_PIXELS;
		if( endPix > nPix ) valid -= endPix - nPix;
	}
	pt->validOwnedPixels = valid;
	pt->width = width;
	pt->height = height;
	if( pt->fbF32.alloc( pt->ownedPixels * sizeof( float4 ) ) ) return 1;
	if( pt->fbU8.alloc( pt->ownedPixels * sizeof( uchar4 ) ) ) return 1;
	if( allocWork( pt ) ) return 1;
	return mvrt_pt_clear_framebuffer( pt, stream ); // :88
}

MVRT_EXPORT int mvrt_pt_load_hdri( mvrt_pt* pt, void* stream, const float* rgbaHost, int width, int height, const float* rgbaPrimaryHost, int widthPrimary,
								   int heightPrimary )
{
	REQUIRE( pt && rgbaHost && width > 0 && height > 0, "mvrt_pt_load_hdri: bad image" );
	if( pt->drain() ) return 1;
	hipStream_t st = (hipStream_t)stream;
	const uint64_t n = (uint64_t)width * height;
	if( pt->hdriPixels.alloc( n * 16 ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( pt->hdriPixels.p, rgbaHost, n * 16, hipMemcpyHostToDevice, st ) );
	DevBuf satF64;
	if( satF64.alloc( n * 8 ) ) return 1;
	// renderCommon.hpp:243-311: uniform table, then one cosine-weighted table per axis
	const f3 axes[6] = { mk3( 1, 0, 0 ), mk3( -1, 0, 0 ), mk3( 0, 1, 0 ), mk3( 0, -1, 0 ), mk3( 0, 0, 1 ), mk3( 0, 0, -1 ) };
	for( int i = 0; i < 7; i++ )
	{
		if( pt->hdriSat[i].alloc( n * 4 ) ) return 1;
		if( launchHdriSat( pt->hdriPixels.as<float4>(), width, height, satF64.as<double>(), pt->hdriSat[i].as<uint32_t>(), i > 0, i > 0 ? axes[i - 1] : mk3( 0, 0, 0 ), st ) )
			return 1;
	}
	pt->hdri.pixels = pt->hdriPixels.as<float4>();
	pt->hdri.sat = pt->hdriSat[0].as<uint32_t>();
	for( int i = 0; i < 6; i++ ) pt->hdri.sats[i] = pt->hdriSat[i + 1].as<uint32_t>();
	pt->hdri.width = width;
	pt->hdri.height = height;
	if( rgbaPrimaryHost ) // HDRI::loadPrimary, :315-326
	{
		const uint64_t np = (uint64_t)widthPrimary * heightPrimary;
		if( pt->hdriPrimary.alloc( np * 16 ) ) return 1;
		MVRT_HIP( hipMemcpyAsync( pt->hdriPrimary.p, rgbaPrimaryHost, np * 16, hipMemcpyHostToDevice, st ) );
		pt->hdri.pixelsPrimary = pt->hdriPrimary.as<float4>();
		pt->hdri.widthPrimary = widthPrimary;
		pt->hdri.heightPrimary = heightPrimary;
	}
	else
	{
		pt->hdriPrimary.release();
		pt->hdri.pixelsPrimary = nullptr;
		pt->hdri.widthPrimary = width; // see mvrt.h: the reference would index with 0x0 here
		pt->hdri.heightPrimary = height;
	}
	MVRT_HIP( hipStreamSynchronize( st ) ); // :313
	return 0;
}
MVRT_EXPORT int mvrt_pt_load_hdri_file( mvrt_pt* pt, void* stream, const char* file, const char* filePrimary )
{
	std::vector<float> a, b;
	int w = 0, h = 0, wp = 0, hp = 0;
	if( loadRgbe( file, a, &w, &h ) ) return 1;
	if( filePrimary && loadRgbe( filePrimary, b, &wp, &hp ) ) return 1;
	return mvrt_pt_load_hdri( pt, stream, a.data(), w, h, filePrimary ? b.data() : nullptr, wp, hp );
}
// host-only: the decoder behind mvrt_pt_load_hdri_file (pr::Image2DRGBA32::loadFromHDR's role, PathTracer.hpp:106-113)
MVRT_EXPORT int mvrt_rgbe_read_file( const char* file, float* rgbaHost, uint64_t capacityPixels, int* width, int* height )
{
	REQUIRE( file && width && height, "null argument" );
	std::vector<float> a;
	if( loadRgbe( file, a, width, height ) ) return 1;
	if( !rgbaHost ) return 0;
	REQUIRE( (uint64_t)*width * *height <= capacityPixels, "%s: %dx%d pixels do not fit %llu", file, *width, *height, (unsigned long long)capacityPixels );
	memcpy( rgbaHost, a.data(), a.size() * sizeof( float ) );
	return 0;
}
MVRT_EXPORT int mvrt_pt_download_hdri_sat( mvrt_pt* pt, int which, uint32_t* satHost )
{
	REQUIRE( pt && which >= 0 && which < 7 && pt->hdriSat[which].p, "no such HDRI table" );
	MVRT_HIP( hipMemcpy( satHost, pt->hdriSat[which].p, pt->hdriSat[which].bytes, hipMemcpyDeviceToHost ) );
	return 0;
}
MVRT_EXPORT int mvrt_pt_set_hdri_scale( mvrt_pt* pt, float scale )
{
	REQUIRE( pt, "null argument" );
	if( pt->flush() ) return 1; // pending steps are launched with the scale they were issued under (HDRI is a by-value kernel argument)
	pt->hdri.scale = scale;
	return 0;
}
MVRT_EXPORT int mvrt_pt_update_scene( mvrt_pt* pt, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
									  const float origin[3], float dps, int gridRes )
{
	REQUIRE( pt, "null argument" );
	if( pt->drain() ) return 1;
	return mvrt_svo_build( pt->intersector, verticesHost, vcolorsHost, vemissionsHost, nVertices, stream, origin, dps, gridRes );
}
MVRT_EXPORT mvrt_svo* mvrt_pt_intersector( mvrt_pt* pt ) { return pt ? pt->intersector : nullptr; }

MVRT_EXPORT int mvrt_pt_step( mvrt_pt* pt, void* stream, const float camera[15] )
{
	REQUIRE( pt && pt->setupDone, "mvrt_pt_step: call mvrt_pt_setup first" );
	REQUIRE( pt->intersector->nodes, "mvrt_pt_step: no scene (updateScene / upload first)" );
	REQUIRE( pt->fbF32.p, "mvrt_pt_step: no frame buffer (resizeFrameBufferIfNeeded first)" );
	REQUIRE( !( 0.0f < pt->hdri.scale ) || pt->hdri.pixels, "mvrt_pt_step: HDRI enabled (scale > 0) but none loaded" );
	PtFrame f;
	f.width = pt->width;
	f.height = pt->height;
	f.tileIndex = pt->tileIndex;
	f.tileCount = pt->tileCount;
	f.ownedPixels = pt->ownedPixels;
	f.validOwnedPixels = pt->validOwnedPixels;
	(void)f;
	// deferred: remember the camera; the pass is launched when `batch` steps are pending or a consumer joins
	if( pt->pendingCams.empty() )
	{
		pt->pendingIteration = pt->steps;
		pt->pendingStream = (hipStream_t)stream;
	}
	else if( pt->pendingStream != (hipStream_t)stream )
	{
		if( pt->flush() ) return 1;
		pt->pendingIteration = pt->steps;
		pt->pendingStream = (hipStream_t)stream;
	}
	pt->steps++; // PathTracer.hpp:159
	pt->pendingCams.push_back( cameraFrom15( camera ) );
	if( (int)pt->pendingCams.size() >= pt->passSteps() ) return pt->flush( true ); // a full batch: the caller is still stepping
	return 0;
}
int mvrt_pt::flush( bool moreStepsFollow )
{
	if( pendingCams.empty() ) return 0;
	const int n = (int)pendingCams.size();
	// A SMALL pass (a tile share of a multi-GPU frame, a small frame) is dominated by the latency floors of its nine traversal
	// launches and by its un-overlapped shade kernels.  Two sibling passes on two streams, each traversal launch restricted to half
	// of the wave slots, overlap one pass's tails and shading with the other's traversal.  Same per-sample results; the frame-buffer
	// additions stay in step order through the event chain.
	const uint64_t samples = ownedPixels * MVRT_SPP_PER_STEP * (uint64_t)n;
	static const uint64_t splitMax = (uint64_t)mvrtKnob( "MVRT_SPLIT_SMALL_MAX", 40000000ll );
	// ... but only when this pass would otherwise run ALONE: if the caller keeps stepping, or an earlier pass is still in flight, the
	// passes already overlap each other and halving their grids only slows them (measured: 2.68 -> 2.99 ms per step at 16 steps)
	const bool alone = !moreStepsFollow && ( !lastAccum || !pendingJoin || hipEventQuery( lastAccum ) == hipSuccess );
	const bool split = splitSmallPasses && alone && depth >= 2 && n >= 2 && samples <= splitMax;
	std::vector<CameraPinhole> cams( pendingCams );
	const int first = pendingIteration;
	pendingCams.clear();
	if( !split ) return launchPass( cams.data(), first, n, 1 );
	// `ways` sibling passes (at most one per work-buffer slot and per step), each restricted to 1/ways of the wave slots
	static const int envWays = (int)mvrtKnob( "MVRT_SPLIT_WAYS", 2 );
	int ways = envWays < 2 ? 2 : envWays;
	if( ways > depth ) ways = depth;
	if( ways > n ) ways = n;
	int done = 0;
	for( int k = 0; k < ways; k++ )
	{
		const int cnt = ( n - done ) / ( ways - k );
		// (r02: each sibling's traversal launches took 1 / ways of the wave slots; r03 re-measured with the cheaper-to-drain kernel: full grids are 1-2 %
		//  better on a 1/8 share -- dragon 2.744 -> 2.698, rtcamp 2.421 -> 2.393 ms per step -- the dispatcher hands the slots of retiring waves to the sibling)
		static const int siblingDiv = (int)mvrtKnob( "MVRT_SIBLING_GRID_DIV", 1 );
		if( launchPass( cams.data() + done, first + done, cnt, siblingDiv > 0 ? siblingDiv : ways ) ) return 1;
		done += cnt;
	}
	return 0;
}
int mvrt_pt::launchPass( const CameraPinhole* passCams, int iteration, int nSteps, int traceGridDiv )
{
	mvrt_pt* pt = this;
	PtFrame f;
	f.width = width;
	f.height = height;
	f.tileIndex = tileIndex;
	f.tileCount = tileCount;
	f.ownedPixels = ownedPixels;
	f.validOwnedPixels = validOwnedPixels;
	f.iteration = iteration;
	f.nSteps = nSteps;
	f.traceGridDiv = traceGridDiv;
	f.useHints = originHints ? 1 : 0;
	CameraPinhole cams[MVRT_MAX_BATCH];
	for( int b = 0; b < f.nSteps; b++ ) cams[b] = passCams[b];
	hipStream_t user = pendingStream;
	mvrt_pt::Slot& sl = pt->slots[pt->nextSlot];
	pt->lastSlot = pt->nextSlot;
	pt->nextSlot = ( pt->nextSlot + 1 ) % pt->depth;
	REQUIRE( sl.buf.cap >= pt->validOwnedPixels * MVRT_SPP_PER_STEP * f.nSteps, "internal: work buffers not allocated" );
	if( ensureSlotWorkspace( pt, sl ) ) return 1;
	sl.buf.dbgTasks = nullptr;
	if( pt->debugCapture )
	{
		if( sl.dbg.bytes < sl.buf.cap * 4 * MVRT_MAX_DEPTH && sl.dbg.alloc( sl.buf.cap * 4 * MVRT_MAX_DEPTH ) ) return 1;
		sl.buf.dbgTasks = sl.dbg.as<uint32_t>();
	}
	hipStream_t run = user;
	hipEvent_t after = nullptr;
	if( pt->depth > 1 )
	{
		run = sl.stream;
		MVRT_HIP( hipEventRecord( pt->forkEv, user ) ); // everything the caller queued so far (clear, upload, ...) comes first
		MVRT_HIP( hipStreamWaitEvent( run, pt->forkEv, 0 ) );
		after = pt->lastAccum;
	}
	int rc = launchPtStep( pt->intersector->dev(), sl.ws, pt->hdri, pt->pmj.as<float2>(), cams, f, sl.buf, pt->fbF32.as<float4>(), pt->numCUs,
						   pt->profiling ? &pt->prof : nullptr, run, after );
	if( rc ) return rc;
	MVRT_HIP( hipEventRecord( sl.accumDone, run ) );
	pt->lastAccum = sl.accumDone;
	if( pt->depth > 1 ) pt->pendingJoin = true;
	return 0; // profiling events are collected lazily by mvrt_pt_get_stats (no sync inside step)
}
MVRT_EXPORT int mvrt_pt_step_matrices( mvrt_pt* pt, void* stream, const float view[16], const float proj[16], float focus, float lensR )
{
	float cam[15];
	mvrt_camera_from_matrices( view, proj, focus, lensR, cam );
	return mvrt_pt_step( pt, stream, cam );
}
MVRT_EXPORT int mvrt_pt_resolve( mvrt_pt* pt, void* stream )
{
	REQUIRE( pt && pt->fbF32.p, "no frame buffer" );
	if( pt->join( (hipStream_t)stream ) ) return 1;
	return launchResolve( pt->fbF32.as<float4>(), pt->validOwnedPixels, pt->fbU8.as<uchar4>(), (hipStream_t)stream );
}
MVRT_EXPORT int mvrt_pt_to_image_async( mvrt_pt* pt, void* stream, uint8_t* rgbaHost )
{
	if( mvrt_pt_resolve( pt, stream ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( rgbaHost, pt->fbU8.p, pt->validOwnedPixels * 4, hipMemcpyDeviceToHost, (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT int mvrt_pt_get_steps( const mvrt_pt* pt ) { return pt ? pt->steps : 0; }
MVRT_EXPORT uint64_t mvrt_pt_get_number_of_voxels( const mvrt_pt* pt ) { return pt ? pt->intersector->info.numberOfVoxels : 0; }
MVRT_EXPORT uint64_t mvrt_pt_get_octree_bytes( const mvrt_pt* pt ) { return pt ? (uint64_t)pt->intersector->info.numberOfNodes * 68 : 0; }
MVRT_EXPORT int mvrt_pt_read_framebuffer( mvrt_pt* pt, void* stream, float* rgbaHost )
{
	REQUIRE( pt && pt->fbF32.p, "no frame buffer" );
	if( pt->join( (hipStream_t)stream ) ) return 1;
	MVRT_HIP( hipMemcpyAsync( rgbaHost, pt->fbF32.p, pt->ownedPixels * 16, hipMemcpyDeviceToHost, (hipStream_t)stream ) );
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	return 0;
}
MVRT_EXPORT float* mvrt_pt_framebuffer_dev( mvrt_pt* pt ) { return pt ? pt->fbF32.as<float>() : nullptr; }
MVRT_EXPORT uint8_t* mvrt_pt_framebuffer_u8_dev( mvrt_pt* pt ) { return pt ? pt->fbU8.as<uint8_t>() : nullptr; }
MVRT_EXPORT const float* mvrt_pt_sample_radiance_dev( mvrt_pt* pt )
{
	if( !pt ) return nullptr;
	(void)pt->drain();
	return pt->slots[pt->lastSlot].buf.Lsx;
}
MVRT_EXPORT int mvrt_pt_read_sample_radiance( mvrt_pt* pt, float* xyzHost, uint64_t nSamples )
{
	REQUIRE( pt && xyzHost, "null argument" );
	if( pt->drain() ) return 1;
	const PtBuffers& b = pt->slots[pt->lastSlot].buf;
	REQUIRE( b.Lsx && nSamples <= b.cap, "no such samples" );
	MVRT_HIP( hipMemcpy( xyzHost, b.Lsx, nSamples * 4, hipMemcpyDeviceToHost ) );
	MVRT_HIP( hipMemcpy( xyzHost + nSamples, b.Lsy, nSamples * 4, hipMemcpyDeviceToHost ) );
	MVRT_HIP( hipMemcpy( xyzHost + 2 * nSamples, b.Lsz, nSamples * 4, hipMemcpyDeviceToHost ) );
	return 0;
}
MVRT_EXPORT int mvrt_pt_join( mvrt_pt* pt, void* stream )
{
	REQUIRE( pt, "null argument" );
	return pt->join( (hipStream_t)stream );
}
MVRT_EXPORT int mvrt_pt_set_batch_steps( mvrt_pt* pt, int maxSteps )
{
	REQUIRE( pt && maxSteps >= 0 && maxSteps <= MVRT_MAX_BATCH, "batch must be 0 (automatic) or 1..%d", MVRT_MAX_BATCH );
	if( pt->drain() ) return 1;
	pt->batch = maxSteps;
	if( pt->fbF32.p ) return allocWork( pt );
	return 0;
}
MVRT_EXPORT int mvrt_pt_set_split_small_passes( mvrt_pt* pt, int enable )
{
	REQUIRE( pt, "null argument" );
	if( pt->drain() ) return 1;
	pt->splitSmallPasses = enable != 0;
	return 0;
}
MVRT_EXPORT int mvrt_pt_set_origin_hints( mvrt_pt* pt, int enable )
{
	REQUIRE( pt, "null argument" );
	if( pt->flush() ) return 1;
	pt->originHints = enable != 0;
	return 0;
}
MVRT_EXPORT int mvrt_pt_set_pipeline_depth( mvrt_pt* pt, int depth )
{
	REQUIRE( pt && depth >= 1 && depth <= 4, "pipeline depth must be 1..4" );
	if( pt->drain() ) return 1;
	pt->depth = pt->depthWanted = depth;
	if( pt->fbF32.p ) return allocWork( pt );
	return 0;
}

MVRT_EXPORT int mvrt_pt_assemble_tiles( const float* gatheredDev, int tileCount, uint64_t rankStridePixels, int width, int height, float* frameDev, void* stream )
{
	REQUIRE( gatheredDev && frameDev && tileCount >= 1, "bad arguments" );
	return launchAssembleTiles( (const float4*)gatheredDev, tileCount, rankStridePixels, width, height, (float4*)frameDev, (hipStream_t)stream );
}
MVRT_EXPORT int mvrt_resolve_buffer( const float* rgbaF32Dev, uint64_t nPixels, uint8_t* rgbaU8Dev, void* stream )
{
	return launchResolve( (const float4*)rgbaF32Dev, nPixels, (uchar4*)rgbaU8Dev, (hipStream_t)stream );
}

MVRT_EXPORT int mvrt_pt_set_debug_capture( mvrt_pt* pt, int enabled )
{
	REQUIRE( pt, "null argument" );
	if( pt->drain() ) return 1;
	pt->debugCapture = enabled != 0;
	if( !enabled )
		for( mvrt_pt::Slot& sl : pt->slots ) sl.dbg.release();
	return 0;
}
MVRT_EXPORT int mvrt_pt_read_debug_stage( mvrt_pt* pt, int stage, uint32_t* tasksHost, uint64_t capacity, uint32_t* survivorsOut )
{
	REQUIRE( pt && stage >= 0 && stage < MVRT_MAX_DEPTH && survivorsOut, "bad arguments" );
	if( pt->drain() ) return 1;
	const mvrt_pt::Slot& sl = pt->slots[pt->lastSlot];
	REQUIRE( sl.buf.dbgTasks, "debug capture was not enabled for the last pass" );
	MVRT_HIP( hipMemcpy( survivorsOut, sl.buf.liveCount + stage + 1, 4, hipMemcpyDeviceToHost ) );
	REQUIRE( *survivorsOut <= capacity && *survivorsOut <= sl.buf.cap, "tasksHost holds %llu entries, stage %d kept %u", (unsigned long long)capacity, stage, *survivorsOut );
	if( tasksHost && *survivorsOut ) MVRT_HIP( hipMemcpy( tasksHost, sl.buf.dbgTasks + (uint64_t)stage * sl.buf.cap, (uint64_t)*survivorsOut * 4, hipMemcpyDeviceToHost ) );
	return 0;
}

MVRT_EXPORT int mvrt_pt_set_test_free_bytes( mvrt_pt* pt, uint64_t bytes )
{
	REQUIRE( pt, "null argument" );
	pt->testFreeBytes = bytes;
	return 0;
}
MVRT_EXPORT int mvrt_pt_set_profiling( mvrt_pt* pt, int enabled )
{
	REQUIRE( pt, "null argument" );
	pt->profiling = enabled != 0;
	return 0;
}
MVRT_EXPORT int mvrt_pt_reset_stats( mvrt_pt* pt )
{
	REQUIRE( pt, "null argument" );
	if( pt->drain() ) return 1;
	if( pt->buf.stats ) MVRT_HIP( hipMemset( pt->buf.stats, 0, 64 * 8 ) );
	pt->prof.collect();
	pt->prof.ms[0] = pt->prof.ms[1] = pt->prof.ms[2] = 0.0;
	pt->prof.traceLaunches = 0;
	return 0;
}
MVRT_EXPORT int mvrt_pt_get_stats( mvrt_pt* pt, void* stream, mvrt_pt_stats* out )
{
	REQUIRE( pt && out, "null argument" );
	memset( out, 0, sizeof( *out ) );
	if( pt->drain() ) return 1;
	MVRT_HIP( hipStreamSynchronize( (hipStream_t)stream ) );
	if( pt->buf.stats )
	{
		unsigned long long s[6];
		MVRT_HIP( hipMemcpy( s, pt->buf.stats, sizeof( s ), hipMemcpyDeviceToHost ) );
		out->rays = s[0];
		out->shadowRays = s[1];
		out->descents = s[2];
		out->shadowDescents = s[3];
		out->hits = s[4];
		out->samples = s[5];
#ifdef MVRT_UTIL_STATS
		if( getenv( "MVRT_PRINT_UTIL" ) ) // (a diagnostic build's tallies: tools/build_variant.sh util -DMVRT_UTIL_STATS)
		{
			unsigned long long u[4];
			MVRT_HIP( hipMemcpy( u, pt->buf.stats + 8, sizeof( u ), hipMemcpyDeviceToHost ) );
			// u[0], u[1]: refill events and the lanes active right after them; u[2], u[3]: wave-iterations of the node-visit loop and
			// the lanes active in them
			fprintf( stderr, "[util] wave-iterations %llu, active-lane-iterations %llu (%.1f%% of lane slots); refill events %llu; rays %llu -> %.2f lane-iterations per ray\n", u[2], u[3],
					 100.0 * u[3] / ( 64.0 * ( u[2] ? u[2] : 1 ) ), u[0], (unsigned long long)s[0], (double)u[3] / (double)( s[0] ? s[0] : 1 ) );
			unsigned long long c[2];
			MVRT_HIP( hipMemcpy( c, pt->buf.stats + 12, sizeof( c ), hipMemcpyDeviceToHost ) );
			fprintf( stderr, "[util] traversal waves: %.1f%% of their shader clocks in refill sections (result flush, ray loads, setup, hint replay)\n", 100.0 * (double)c[0] / (double)( c[1] ? c[1] : 1 ) );
			unsigned long long w[48];
			MVRT_HIP( hipMemcpy( w, pt->buf.stats + 16, sizeof( w ), hipMemcpyDeviceToHost ) );
			for( int k = 0; k <= MVRT_MAX_DEPTH; k++ )
				fprintf( stderr, "[util] stage %d: longest wave %llu iterations, waves %llu, longest ray %llu iterations\n", k, w[k], w[16 + k], w[32 + k] );
		}
#endif
	}
	pt->prof.collect();
	out->traceLaunches = pt->prof.traceLaunches;
	out->traceKernelMs = pt->prof.ms[MVRT_K_TRACE];
	out->shadeKernelMs = pt->prof.ms[MVRT_K_SHADE];
	out->totalKernelMs = pt->prof.ms[0] + pt->prof.ms[1] + pt->prof.ms[2];
	return 0;
}
