/*
 * mvrt.h -- C-ABI of libmvrt_hip.so, the MI355X-native (gfx950) replacement for the
 * reference's GPU hot path: sparse-voxel-octree traversal, wavefront path tracing with
 * stable live-ray compaction, and the host objects that own it.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository root).  Plain pointers and sizes only: no C++ or torch types.
 *
 * Conventions
 *   - all functions return 0 on success, non-zero on failure; mvrt_last_error() gives text.
 *     (The reference returns void and __debugbreak()s / abort()s: hipUtil.hpp:18-22,
 *     IntersectorOctreeGPU.hpp:48-51.  The header-only C++ mirrors in include/mvrt/ (IntersectorOctreeGPU.hpp, PathTracer.hpp) keep the
 *     void signatures and abort() on a non-zero status.)
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are asynchronous on
 *     it unless stated otherwise, exactly like the reference (PathTracer.hpp:150-169).
 *   - "host" / "dev" in a parameter name says where the pointer must live.
 *   - vectors are 3 packed floats; matrices are 16 floats, column-major (glm).
 *   - camera = the 15 floats of CameraPinhole {m_o, m_front, m_up, m_right, m_tanHthetaY,
 *     m_lensR, m_focus} (renderCommon.hpp:77-83).
 *   - OctreeNode = the reference's 68-byte AoS node (voxCommon.hpp:133-138); VoxelAttirb = 8 bytes
 *     {uchar4 color, uchar4 emission} (voxCommon.hpp:121-125).
 */
#ifndef MVRT_H
#define MVRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVRT_MAX_FLOAT 3.402823466e+38F /* "miss" marker for t, vectorMath.hpp:79 */

typedef struct mvrt_svo mvrt_svo; /* IntersectorOctreeGPU (IntersectorOctreeGPU.hpp:21-275) */
typedef struct mvrt_pt mvrt_pt;	  /* PathTracer (PathTracer.hpp:14-170) */

/* ---- runtime ------------------------------------------------------------------------------ */
const char* mvrt_last_error( void );
int mvrt_device_count( int* count );			/* replaces oroGetDeviceCount, voxPTGPU.cpp:16-39 */
int mvrt_set_device( int device );				/* replaces oroCtxCreate/SetCurrent */
int mvrt_device_name( char* buf, int bufLen );	/* voxPTGPU.cpp:215 device name print */
int mvrt_stream_create( void** stream );		/* oroStreamCreate, voxPTGPU.cpp:41 */
int mvrt_stream_destroy( void* stream );
int mvrt_stream_synchronize( void* stream );	/* oroStreamSynchronize, voxPTGPU.cpp:194 */
int mvrt_device_synchronize( void );
/* plain device buffers for callers without their own allocator (hipUtil.hpp:48-74 Buffer) */
int mvrt_malloc( void** dev, uint64_t bytes );
int mvrt_free( void* dev );
int mvrt_memcpy_h2d( void* dev, const void* host, uint64_t bytes, void* stream );
int mvrt_memcpy_d2h( void* host, const void* dev, uint64_t bytes, void* stream );
int mvrt_memcpy_d2d( void* dstDev, const void* srcDev, uint64_t bytes, void* stream ); /* asynchronous on stream */

/* ---- IntersectorOctreeGPU ------------------------------------------------------------------ */
typedef struct mvrt_svo_info
{
	uint32_t numberOfNodes;	 /* m_numberOfNodes  (IntersectorOctreeGPU.hpp:267) */
	uint32_t numberOfVoxels; /* m_numberOfVoxels (:268) */
	float lower[3];			 /* m_lower (:269) */
	float upper[3];			 /* m_upper (:270) */
	float dps;				 /* m_dps (:271) */
	float emissionScale;	 /* m_emissionScale = 7.5 (:273) */
	uint32_t hasEmission;	 /* m_hasEmission (:274) */
	uint32_t embeddedMask;	 /* 1: child pointers carry the child's mask in bits 24-31 (voxCommon.hpp:7-9) */
	uint32_t gridRes;
	uint32_t levels;		   /* log2(gridRes) = maximum traversal stack depth */
	uint64_t totalDumpedVoxels; /* voxels emitted before de-duplication (build only, else 0) */
	uint32_t flavour;			/* layout behind mvrt_svo_node_buffer_dev: MVRT_FLAVOUR_* */
	uint32_t reserved;
} mvrt_svo_info;
#define MVRT_FLAVOUR_EMBEDDED 0 /* 64-byte lines {children[8] with the child's mask in bits 24-31, nVoxelsPSum[8]} */
#define MVRT_FLAVOUR_PLAIN 1	/* 64-byte lines {children[8], the 8 child masks in 2 words, 6 unused words}; nVoxelsPSum in a separate array */
#define MVRT_FLAVOUR_TREE 2		/* 16-byte two-level bricks {u8 childMask[8]; u32 ownMask; u32 base} (four per 64-byte line) */

int mvrt_svo_create( mvrt_svo** out );
int mvrt_svo_destroy( mvrt_svo* svo ); /* IntersectorOctreeGPU::cleanUp, :26-38 */

/* IntersectorOctreeGPU::build (:40-241): voxelize triangles (six-separating), sort, de-duplicate with
 * integer-mean attributes, build the octree DAG bottom-up and embed child masks -- all on the GPU.
 * vertices/vcolors/vemissions: nVertices*3 host floats (nVertices = 3 * triangles); vcolors/vemissions may
 * be NULL (white / black, voxUtil.hpp:49-61).  gridRes must be a power of two (:48-51).  Blocks until done,
 * like the reference (5 host syncs, :92-211).  Node numbering is deterministic and equals
 * buildOctreeDAGReference's creation order (IntersectorOctree.hpp:11-123); root = last node (:250). */
int mvrt_svo_build( mvrt_svo* svo, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
					const float origin[3], float dps, int gridRes );

/* Same with options.  MVRT_BUILD_NO_DAG: every sibling group becomes a node (the reference with ENABLE_GPU_DAG off,
 * voxKernel.cu:322-334; node numbering = deterministic group order).  MVRT_BUILD_NO_EMBEDDED_MASK: child pointers stay
 * plain indices and a node's mask is read from the node (voxCommon.hpp:353-356); chosen automatically when the octree has
 * >= 0xFFFFFF nodes, the limit of the embedded form (IntersectorOctreeGPU.hpp:231).  MVRT_BUILD_CONSERVATIVE: conservative
 * voxelization -- every voxel a triangle touches -- instead of the six-separating one (VTContext's sixSeparating == false,
 * voxelization.hpp:186-189,296-301; the reference's GPU build hard-codes six-separating, voxKernel.cu:68,109, its CPU demo has the
 * switch, voxRT.cpp:107,389). */
#define MVRT_BUILD_NO_DAG 1
#define MVRT_BUILD_NO_EMBEDDED_MASK 2
#define MVRT_BUILD_CONSERVATIVE 4
int mvrt_svo_build_ex( mvrt_svo* svo, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
					   const float origin[3], float dps, int gridRes, int flags );
/* Seeded synthetic octree for HBM-bound stress runs (BASELINE.json configs[4]): nRandomVoxels uniformly random cells of the
 * gridRes^3 grid (duplicates merge), hash-derived colours, ~1/256 emissive; generated, sorted and built on the GPU.
 * voxel i: h = splitmix64(seed + i); x = h & (res-1), y = (h >> 21) & (res-1), z = (h >> 42) & (res-1); c = splitmix64(h):
 * colour = (c & 0xFFFFFF) | 0x404040, emission = colour if (c >> 56) == 0 else 0. */
int mvrt_svo_build_synthetic( mvrt_svo* svo, int gridRes, uint64_t nRandomVoxels, uint64_t seed, const float origin[3], float dps, int flags, void* stream );

/* Adopt an SVO built elsewhere (e.g. IntersectorOctree::buildDAGReference on the CPU, IntersectorOctree.hpp:
 * 224-231): nodes in the reference's 68-byte layout, root last.  embeddedMask = 0 selects the variant where
 * the mask is fetched from the node (voxCommon.hpp:353-356; required above 0xFFFFFF nodes). */
/* nVoxelsPSum is used as stored: vIndex = the sum of the stored values along the path (voxCommon.hpp:388-391), whatever they are -- e.g. all zero
 * for buildOctreeNaive's nodes (IntersectorOctree.hpp:195), which then give vIndex 0 like the reference.  (Internally the last level's value is
 * replaced by a popcount of the mask when the upload is found to carry exactly that there, as every octree built by bottomUpOctreeBuild does.) */
int mvrt_svo_upload( mvrt_svo* svo, const void* nodes68Host, uint32_t numberOfNodes, const void* attribs8Host, uint32_t numberOfVoxels, const float origin[3],
					 float dps, int gridRes, int hasEmission, int embeddedMask, void* stream );
int mvrt_svo_get_info( const mvrt_svo* svo, mvrt_svo_info* info );
int mvrt_svo_set_emission_scale( mvrt_svo* svo, float scale ); /* m_emissionScale (:273) */
/* bytes of the device structure the traversal and mvrt_svo_download work from (the reference's layout would be numberOfNodes * 68):
 * 64-byte lines per node (+ a 16 MiB prefix table, the 32-byte children arrays and prefix tables the traversal reads and, for octrees built here, the cell index
 * that turns a hit voxel's path into its index: ~25 bytes per voxel) for DAG octrees; for GPU-built octrees WITHOUT node sharing whose masks are not embedded
 * ("tree" flavour: MVRT_BUILD_NO_DAG with >= 0xFFFFFF nodes or MVRT_BUILD_NO_EMBEDDED_MASK) 5 bytes per node + one 16-byte two-level brick
 * per node of every second level (four bricks share a 64-byte line). */
uint64_t mvrt_svo_traversal_bytes( const mvrt_svo* svo );
/* m_nodeBuffer / m_vAttributeBuffer (:265-266): the device arrays.  Attributes are the reference's VoxelAttirb[numberOfVoxels]; nodes are
 * this library's 64-byte lines {u32 children[8]; u32 nVoxelsPSum[8]} (the reference's 68-byte node minus its leading mask word, which rides
 * in bits 24-31 of the parent's pointer); for an octree too large for that (no node sharing, > 2^24 nodes: the tree flavour) they are its 16-byte
 * two-level bricks {u8 childMask[8]; u32 ownMask; u32 base}.  mvrt_svo_info::flavour says which of the three layouts the pointer has; use
 * mvrt_svo_download for the reference layout. */
const void* mvrt_svo_node_buffer_dev( const mvrt_svo* svo );
const void* mvrt_svo_attribute_buffer_dev( const mvrt_svo* svo );
/* read the SVO back in the reference layout (parity checks of build); either pointer may be NULL */
int mvrt_svo_download( const mvrt_svo* svo, void* nodes68Host, void* attribs8Host, uint64_t* mortonHost, void* stream );

/* Host-callable batch form of the device method IntersectorOctreeGPU::intersect (:243-251) ==
 * octreeTraverse_EfficientParametric (voxCommon.hpp:231-423).  SoA device arrays of n floats each.
 * isShadowDev: per-ray flags (nonzero = isShadowRay: vIndex not accumulated) or NULL for "all false".
 * Outputs: t (MVRT_MAX_FLOAT on a miss), nMajor (1:x 2:y 0:z; -1 on a miss), vIndex (0 on a miss / shadow);
 * descentsDev (optional) = child-pointer fetches per ray (voxCommon.hpp:381), the unit of the
 * algorithmic-bytes model in DESIGN.md. */
int mvrt_trace_batch( const mvrt_svo* svo, uint64_t n, const float* roxDev, const float* royDev, const float* rozDev, const float* rdxDev, const float* rdyDev,
					  const float* rdzDev, const uint8_t* isShadowDev, float* tDev, int32_t* nMajorDev, uint32_t* vIndexDev, uint32_t* descentsDev, void* stream );
/* Same with a START HINT per ray (new; no counterpart in the reference, which starts every ray at the root, voxCommon.hpp:306-312):
 * originVoxelMortonDev[i] = the 3-bits-per-level root->voxel path (= Morton code of the grid cell, x = bit 0) of ANY voxel that exists in the
 * octree, or ~0 for "no hint".  The traversal replays the walk from the root along that path arithmetically for as long as the ray's origin lies
 * in the same octants and starts below the root; every output, descents included, is identical to the unhinted call for every valid hint -- a
 * hint near the origin only makes it cheaper (the path tracer hints each secondary ray with the voxel its path just hit).  Embedded-mask
 * octrees only; ignored otherwise.  A code that names a cell WITHOUT a voxel is an error the library does not detect. */
int mvrt_trace_batch_hinted( const mvrt_svo* svo, uint64_t n, const float* roxDev, const float* royDev, const float* rozDev, const float* rdxDev, const float* rdyDev,
							 const float* rdzDev, const uint8_t* isShadowDev, const uint64_t* originVoxelMortonDev, float* tDev, int32_t* nMajorDev, uint32_t* vIndexDev,
							 uint32_t* descentsDev, void* stream );
/* convenience: packed host arrays (n*3 floats), synchronous */
int mvrt_trace_batch_host( const mvrt_svo* svo, uint64_t n, const float* roHost, const float* rdHost, const uint8_t* isShadowHost, float* tHost, int32_t* nMajorHost,
						   uint32_t* vIndexHost, uint32_t* descentsHost );

/* The `render` kernel (voxKernel.cu:437-483) as launched by voxRTGPU.cpp:191-203: one primary ray per pixel
 * through the pixel centre; colour = voxel colour (showVertexColor) or the hit normal.  rgbaDev: width*height
 * uchar4.  Optional per-pixel outputs for parity checks (may be NULL). */
int mvrt_render_primary( const mvrt_svo* svo, const float camera[15], int width, int height, int showVertexColor, uint8_t* rgbaDev, float* tDev, int32_t* nMajorDev,
						 uint32_t* vIndexDev, uint32_t* descentsDev, void* stream );

/* CameraPinhole::initFromPerspective (renderCommon.hpp:21-35) */
int mvrt_camera_from_matrices( const float view[16], const float proj[16], float focus, float lensR, float cameraOut[15] );

/* Stable stream compaction (StreamCompaction::filter semantics, StreamCompaction.hpp:87-184): for n device
 * flags, dstIndexDev[i] = number of kept items before i (0xFFFFFFFF if dropped), *keptDev = kept count.
 * Same wave64 ballot + ordered block scan the path tracer uses for live rays. */
int mvrt_compact_indices( const uint8_t* keepDev, uint64_t n, uint32_t* dstIndexDev, uint32_t* keptDev, void* stream );

/* ---- PathTracer ---------------------------------------------------------------------------- */
int mvrt_pt_create( mvrt_pt** out );
int mvrt_pt_destroy( mvrt_pt* pt );				/* PathTracer::cleanUp, PathTracer.hpp:71-79 */
/* PathTracer::setup (:43-69): PMJ02 table (pmjSampler.hpp:114-144) and work buffers.  The reference's kernel
 * path / include dir / isNvidia arguments have no meaning here (no runtime compilation). */
int mvrt_pt_setup( mvrt_pt* pt, void* stream );
/* read back PMJSampler::m_samples (pmjSampler.hpp:114-144): 128 sequences x 4096 float2 = 4 MiB */
int mvrt_pt_download_pmj( mvrt_pt* pt, float* tableHost );
int mvrt_pt_resize_framebuffer_if_needed( mvrt_pt* pt, void* stream, int width, int height ); /* :81-97 */
int mvrt_pt_clear_framebuffer( mvrt_pt* pt, void* stream );									 /* :98-102, steps = 0 */
/* PathTracer::loadHDRI (:104-116) + HDRI::load/loadPrimary (renderCommon.hpp:214-326): decoded float4 pixels.
 * rgbaPrimaryHost may be NULL (then primary lookups use the lighting map's pixels AND size -- the reference
 * would read out of bounds, renderCommon.hpp:356-363). */
int mvrt_pt_load_hdri( mvrt_pt* pt, void* stream, const float* rgbaHost, int width, int height, const float* rgbaPrimaryHost, int widthPrimary, int heightPrimary );
/* same from Radiance .hdr files (RGBE, flat or RLE; value = c * 2^(E-136)); filePrimary may be NULL */
int mvrt_pt_load_hdri_file( mvrt_pt* pt, void* stream, const char* file, const char* filePrimary );
/* host-only: decode a Radiance .hdr file to float4 pixels (alpha 1) exactly as mvrt_pt_load_hdri_file does -- flat and new-RLE scanlines,
 * value = c * 2^(E-136).  rgbaHost may be NULL to query the size. */
int mvrt_rgbe_read_file( const char* file, float* rgbaHost, uint64_t capacityPixels, int* width, int* height );
/* read back one importance table (parity checks): which = 0 uniform, 1..6 = +x,-x,+y,-y,+z,-z; width*height u32 */
int mvrt_pt_download_hdri_sat( mvrt_pt* pt, int which, uint32_t* satHost );
int mvrt_pt_set_hdri_scale( mvrt_pt* pt, float scale ); /* HDRI::m_scale = 1.75 (renderCommon.hpp:480); <= 0 disables NEE */
/* PathTracer::updateScene (:139-148) -> IntersectorOctreeGPU::build */
int mvrt_pt_update_scene( mvrt_pt* pt, const float* verticesHost, const float* vcolorsHost, const float* vemissionsHost, uint64_t nVertices, void* stream,
						  const float origin[3], float dps, int gridRes );
mvrt_svo* mvrt_pt_intersector( mvrt_pt* pt ); /* &PathTracer::m_intersectorOctreeGPU (:18); owned by pt */
/* PathTracer::step (:150-169): one launch of renderPT semantics = 16 spp for every pixel, iteration = steps++.
 * Wavefront implementation: generate -> [trace -> count/scan -> shade+compact] x <= 9 -> accumulate. */
int mvrt_pt_step( mvrt_pt* pt, void* stream, const float camera[15] );
/* same with the matrices GetCameraMatrix produces in the reference (:152-156) */
int mvrt_pt_step_matrices( mvrt_pt* pt, void* stream, const float view[16], const float proj[16], float focus, float lensR );
/* Consecutive step() calls are pipelined on internal streams (`depth` steps in flight, default 2; 1 = none) so that
 * the thin late bounces of one step overlap the dense early bounces of the next.  Results are unchanged (frame-buffer
 * additions are chained in step order).  Every call that consumes the frame buffer (resolve, to_image, read, clear)
 * first makes `stream` wait for the steps in flight; callers that read mvrt_pt_framebuffer_dev() themselves call
 * mvrt_pt_join( pt, stream ) before. */
int mvrt_pt_set_pipeline_depth( mvrt_pt* pt, int depth );
/* Secondary rays (shadow, extra, bounce) start below the root, hinted with the voxel their path hit last (mvrt_trace_batch_hinted); default on.
 * 0 = every ray walks from the root like the reference's.  Results are identical either way. */
int mvrt_pt_set_origin_hints( mvrt_pt* pt, int enable );
/* step() is DEFERRED: up to maxSteps (1 = launch immediately; 0 = automatic, the default: about two full-HD steps of samples per pass and at most half of the caller's
 * frame -- the steps between its last two clear_framebuffer calls --, so that a frame is at least two passes that overlap) consecutive step() calls are merged into one
 * wavefront pass -- larger launches, identical per-sample results, additions to the frame buffer still step by step.
 * Any consumer (resolve, to_image, read, clear, join, get_stats ...) launches what is pending first. */
int mvrt_pt_set_batch_steps( mvrt_pt* pt, int maxSteps );
/* A SMALL pass (<= 40 M samples: a tile share of a multi-GPU frame, a small frame) of >= 2 merged steps is launched as two sibling
 * passes on two internal streams, so that the launch tails and the shading of one overlap with the traversal of the other
 * (default on; needs pipeline depth >= 2).  Results are unchanged. */
int mvrt_pt_set_split_small_passes( mvrt_pt* pt, int enable );
int mvrt_pt_join( mvrt_pt* pt, void* stream );
int mvrt_pt_resolve( mvrt_pt* pt, void* stream );						/* :130-137, renderResolve */
int mvrt_pt_to_image_async( mvrt_pt* pt, void* stream, uint8_t* rgbaHost ); /* :118-129 resolve + DtoH (caller syncs) */
int mvrt_pt_get_steps( const mvrt_pt* pt );								/* :33 */
uint64_t mvrt_pt_get_number_of_voxels( const mvrt_pt* pt );				/* :34-37 */
uint64_t mvrt_pt_get_octree_bytes( const mvrt_pt* pt );					/* :38-41 (nodes * 68) */
/* F32 accumulation buffer (m_frameBufferF32, :22): float4 per OWNED pixel, xyz = sum, w = spp */
int mvrt_pt_read_framebuffer( mvrt_pt* pt, void* stream, float* rgbaHost /* ownedPixels*4 */ );
float* mvrt_pt_framebuffer_dev( mvrt_pt* pt );
uint8_t* mvrt_pt_framebuffer_u8_dev( mvrt_pt* pt );

/* Multi-GPU tile split (new; the reference has no multi-GPU path).  The frame is cut into the reference's own
 * 256-pixel blocks (RENDER_NUMBER_OF_THREAD, renderCommon.hpp:13) dealt round-robin: this handle renders blocks
 * b with b % tileCount == tileIndex.  Owned pixels are stored compactly in block order.  Call before
 * resize_framebuffer.  Samples depend only on (global pixel index, spp), so any split reproduces the 1-GPU image. */
int mvrt_pt_set_tile( mvrt_pt* pt, int tileIndex, int tileCount );
uint64_t mvrt_pt_owned_pixels( const mvrt_pt* pt ); /* padded to whole 256-pixel blocks */
/* scatter gathered per-rank buffers (rank-major, each rankStridePixels float4) back to a width*height frame */
int mvrt_pt_assemble_tiles( const float* gatheredDev, int tileCount, uint64_t rankStridePixels, int width, int height, float* frameDev, void* stream );
/* renderResolve (voxKernel.cu:779-795) on an arbitrary float4 buffer */
int mvrt_resolve_buffer( const float* rgbaF32Dev, uint64_t nPixels, uint8_t* rgbaU8Dev, void* stream );

/* Per-sample radiance of the LAST step (debug / parity): ownedPixels*16*3 floats on the device */
const float* mvrt_pt_sample_radiance_dev( mvrt_pt* pt );
/* host copy of the x, y, z planes (nSamples floats each) of the last pass; sample = (step * pixels + pixel) * 16 + spp */
int mvrt_pt_read_sample_radiance( mvrt_pt* pt, float* xyzHost, uint64_t nSamples );

/* Debug capture (parity of the live-path compaction, StreamCompaction.hpp:87-184 semantics): when enabled, every shade stage of a pass also
 * keeps a copy of the survivor list it wrote.  read_debug_stage returns, for the LAST pass, the sample ids ("tasks": ((step * pixels + pixel)
 * * 16 + spp)) of the paths that survived `stage` (0 = primary .. 7), in the order of their compacted slots: stable compaction <=> ascending. */
int mvrt_pt_set_debug_capture( mvrt_pt* pt, int enabled );
int mvrt_pt_read_debug_stage( mvrt_pt* pt, int stage, uint32_t* tasksHost, uint64_t capacity, uint32_t* survivorsOut );

/* Counters and timings of the work since the last reset (all steps). */
typedef struct mvrt_pt_stats
{
	uint64_t samples;		 /* paths started */
	uint64_t rays;			 /* intersect() calls (primary + shadow + extra + bounce) */
	uint64_t shadowRays;	 /* of which isShadowRay */
	uint64_t descents;		 /* child fetches of non-shadow rays */
	uint64_t shadowDescents; /* child fetches of shadow rays */
	uint64_t hits;			 /* non-shadow rays that hit */
	uint64_t traceLaunches;	 /* launches of the traversal kernel */
	double traceKernelMs;	 /* summed HIP-event time of the traversal kernel (profiling on) */
	double shadeKernelMs;	 /* summed time of shade+compact kernels (profiling on) */
	double totalKernelMs;	 /* summed time of every kernel of step() (profiling on) */
} mvrt_pt_stats;
/* Failure-path testing: when bytes != 0 the path-state budget of resize / set_pipeline_depth / set_batch_steps is computed against this much "free HBM"
 * instead of what hipMemGetInfo reports.  After a failed (re)allocation the handle has NO frame (steps fail with an error until the next
 * successful mvrt_pt_resize_framebuffer_if_needed). */
int mvrt_pt_set_test_free_bytes( mvrt_pt* pt, uint64_t bytes );
int mvrt_pt_set_profiling( mvrt_pt* pt, int enabled ); /* HIP events on `stream` around each kernel of step(); read by get_stats */
int mvrt_pt_reset_stats( mvrt_pt* pt );
int mvrt_pt_get_stats( mvrt_pt* pt, void* stream, mvrt_pt_stats* out ); /* synchronises the stream */

#ifdef __cplusplus
}
#endif
#endif /* MVRT_H */
