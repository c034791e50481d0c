// launch.h -- host-side launch wrappers implemented in the .hip translation units.
#pragma once
#include "mvrt_common.h"

// ---- wavefront path-tracer work buffers (all device pointers, capacity `cap` paths) ---------------
struct PathSet // ping-ponged between shade stages (compacted by the writer)
{
	uint32_t* task;		   // local sample id = ownedPixel * 16 + localSpp
	uint32_t* org;		   // stage k>0: start-below-the-root hint = leading levels of the path of the voxel the ray starts on (traverse_stream.h)
	float *rox, *roy, *roz; // ray origin (stage k>0: the hit point the bounce leaves from)
	float *rdx, *rdy, *rdz; // ray direction (stage k>0: the Lambert bounce direction)
	float *Tx, *Ty, *Tz;	// throughput after the last T *= R
	float *Lx, *Ly, *Lz;	// radiance accumulated so far
	float *nx, *ny, *nz;	// pending next-event contribution, added if the shadow ray misses
};
struct PtBuffers
{
	PathSet set[2];
	float *sx, *sy, *sz; // shadow-ray direction (kind 1), origin = ro
	float *ex, *ey, *ez; // extra Lambert ray direction (kind 2, stage 1 only), origin = ro
	// hit records written by the traversal kernel, one array per ray kind
	float* hitT;	  // kind 0: t (MAX_FLOAT = miss)
	uint8_t* hitN;	  // kind 0: nMajor
	uint8_t* hitS;	  // kind 1: 1 if the shadow ray is occluded
	uint8_t* hitE;	  // kind 2: 1 if the extra ray hit
	uint64_t* hitPath;	// kind 0: voxel path (persistent traversal; vIndex is derived by the shade kernel)
	uint64_t* hitEPath; // kind 2: voxel path
	float *Lsx, *Lsy, *Lsz; // final radiance per sample, indexed by task (read by accumulate)
	uint32_t* blockCount; // survivors per 256-path virtual block; exclusive-scanned in place
	uint32_t* liveCount;  // [stage] number of live paths entering stage k (0..9)
	unsigned long long* cursors; // [16] per-stage ray cursors of the persistent traversal waves (zeroed by generate)
	unsigned long long* stats; // [0] rays [1] shadowRays [2] descents [3] shadowDescents [4] hits [5] samples
	uint64_t cap;
	uint32_t* dbgTasks; // debug capture (host-side use only): [MVRT_MAX_DEPTH][cap] task ids of the survivors each shade stage wrote, in slot order
	const PtBuffers* selfDev; // device-resident copy of this table (read with scalar loads where needed, see PtIO)
};

struct PtFrame
{
	int width, height;
	int tileIndex, tileCount;
	uint64_t ownedPixels;	   // padded to whole 256-pixel blocks
	uint64_t validOwnedPixels; // pixels that exist in the image
	int iteration;			   // iteration of the first step of this batch
	int nSteps;				   // consecutive step() calls merged into this wavefront pass (1..MVRT_MAX_BATCH)
	int traceGridDiv;		   // >1: the traversal launches of this pass take only 1/div of the wave slots (it shares the GPU with a sibling pass)
	int useHints;			   // secondary rays start below the root
};
#define MVRT_MAX_BATCH 8

// scratch of the persistent traversal kernels (traverse_stream.h): HBM spill rows [level][lane] + a ray cursor.
// One workspace serves one in-flight launch: calls that share it must be ordered on one stream.
struct TraceWorkspace
{
	uint4* spill;
	uint64_t spillStride; // lanes = CUs * 32 waves * 64
	unsigned long long* cursor;
	uint64_t* paths; // per-ray voxel paths of batch / primary-cast launches, resolved to vIndex by a dense pass
	uint64_t pathCap;
	uint32_t* spillMask; // non-embedded flavour: node masks of evicted stack entries, [level][lane]
	uint32_t* spillMask2; // tree flavour: the second mask word of evicted entries
};
uint64_t traceWorkspaceLanes();

enum MvrtKernelClass
{
	MVRT_K_TRACE = 0,
	MVRT_K_SHADE = 1,
	MVRT_K_OTHER = 2
};

int launchTraceBatch( const SvoDev& svo, const TraceWorkspace& ws, uint64_t n, const float* rox, const float* roy, const float* roz, const float* rdx, const float* rdy, const float* rdz,
					  const uint8_t* isShadow, float* t, int32_t* nMajor, uint32_t* vIndex, uint32_t* descents, hipStream_t stream, const uint64_t* originPath = nullptr );
int launchRenderPrimary( const SvoDev& svo, const TraceWorkspace& ws, const CameraPinhole& cam, int W, int H, int showVertexColor, uchar4* rgba, float* t, int32_t* nMajor, uint32_t* vIndex,
						 uint32_t* descents, hipStream_t stream );
int launchCompactIndices( const uint8_t* keep, uint64_t n, uint32_t* dstIndex, uint32_t* kept, uint32_t* blockScratch, hipStream_t stream );

// one PathTracer::step().  `mark(class)` is called before/after each kernel when profiling is on.
struct PtProfiler
{
	virtual void begin( int kernelClass, hipStream_t s ) = 0;
	virtual void end( hipStream_t s ) = 0;
};
int launchPtStep( const SvoDev& svo, const TraceWorkspace& ws, const HdriDev& hdri, const float2* pmj, const CameraPinhole* cams /* frame.nSteps */, const PtFrame& frame, const PtBuffers& buf, float4* frameBuffer,
				  int numCUs, PtProfiler* prof, hipStream_t stream, hipEvent_t accumulateAfter );

int launchResolve( const float4* fb, uint64_t n, uchar4* out, hipStream_t stream );
int launchAssembleTiles( const float4* gathered, int tileCount, uint64_t rankStridePixels, int W, int H, float4* frame, hipStream_t stream );
// nonEmbedded != 0: psum goes to psumCold (nNodes * 8 u32) and the 8 child masks into Node64::psum[0..1]
int launchConvertNodes( const uint8_t* nodes68, uint32_t nNodes, Node64* out, uint8_t* masks, uint32_t* psumCold, int nonEmbedded, hipStream_t stream );
int launchNodesTo68( const Node64* nodes, const uint8_t* masks, const uint32_t* psumCold, uint32_t nNodes, uint8_t* nodes68, int nonEmbedded, hipStream_t stream );
int launchCheckLeafPsum( const Node64* nodes, const uint8_t* masks, uint32_t nNodes, uint32_t* badDev, hipStream_t stream ); // embedded flavour, after an upload
int launchSplitPsum( Node64* nodes, const uint8_t* masks, uint32_t* psumCold, uint64_t nNodes, hipStream_t stream ); // in place, after a build
// tree flavour -> the reference's 68-byte nodes (mask, children[8], nVoxelsPSum[8]) from { mask, first child } per node
int launchTreeTo68( const uint8_t* masks, const uint32_t* first, const uint32_t* levelBase, const uint32_t* levelCount, int levels, uint32_t nNodes, uint32_t nVoxels, uint8_t* nodes68,
					hipStream_t stream );
int launchCopyKids( const Node64* nodes, uint64_t nNodes, uint32_t* kids, hipStream_t stream ); // embedded flavour: compact children array for the traversal
int launchBuildTopTable( const Node64* nodes, uint32_t rootIndex, uint32_t k, uint2* table, hipStream_t stream ); // embedded flavour only
// embedded flavour: node reference (index | mask << 24) per path prefix of 0..tabLevels levels, level l at prefixTabOffset( l ) (traverse_stream.h)
int launchBuildPrefixRefs( const uint32_t* kids, uint32_t rootRef, uint32_t tabLevels, uint32_t* table, hipStream_t stream );
// cell index of a build (SvoDev::cellBlocks / cellEntries) from its sorted voxel codes: number the occupied blocks (blocks[] preset to ~0, counter to 0), then fill the entries (zeroed)
int launchNumberCellBlocks( const uint64_t* morton, uint64_t n, uint32_t cellBits, uint32_t* blocks, uint32_t* counterDev, hipStream_t stream );
int launchFillCellIndex( const uint64_t* morton, uint64_t n, uint32_t cellBits, const uint32_t* blocks, uint2* entries, hipStream_t stream );
int launchHdriSat( const float4* pixels, int w, int h, double* satF64, uint32_t* satOut, int cosWeighted, f3 axis, hipStream_t stream );

// GPU SVO construction (svo_build.hip)
struct SvoBuildResult
{
	Node64* nodes;
	uint8_t* masks;
	uint32_t* psumCold; // non-embedded flavour only
	uint2* attrs;
	uint64_t* morton; // kept for parity checks (sorted unique codes)
	uint32_t nNodes, nVoxels, hasEmission, embedded;
	uint64_t totalDumped;
	// "tree" flavour (no DAG, masks not embedded): `nodes` holds nBricks two-level bricks, `masks` the per-node masks and `treeFirst` the
	// per-node first-child index (reference numbering); node ranges per builder level (0 = parents of voxels)
	uint32_t tree, nBricks, treeRoot;
	uint32_t* treeFirst;
	uint32_t treeLevelBase[24], treeLevelCount[24];
};
// flags: 1 = no DAG de-duplication (every sibling group is a node), 2 = never embed masks in child pointers
int svoBuildFromTriangles( const float* vertsHost, const float* colsHost, const float* emisHost, uint64_t nVertices, f3 origin, float dps, int gridRes, int flags,
						   hipStream_t stream, SvoBuildResult* out );
int svoBuildSynthetic( uint64_t nRandomVoxels, uint64_t seed, int gridRes, int flags, hipStream_t stream, SvoBuildResult* out );
