// traverse_stream.h -- persistent-wave octree traversal for gfx950.
//
// Three node-reference flavours (template parameter FL; EMBED = FL == 0, TREE = FL == 2):
//   FL = 2 (tree) : octrees WITHOUT node sharing whose masks cannot be embedded (the HBM-resident 8192^3 stress octree).  The children of a
//                   node are consecutive nodes, so a child reference is (first child, mask) and the traversal reads two-level BRICKS: a
//                   16-byte record per node of every second level { u8 childMask[8]; u32 ownMask; u32 base } (four share a 64-byte line).
//                   A lane sitting on a brick root keeps the whole record in registers; descending into child c is arithmetic
//                   (base + popcount of the child masks before c), and only the next descent -- onto the grandchild's brick -- is a new
//                   fetch.  One dependent HBM fetch per TWO levels; the hit voxel's index falls out of the last step
//                   (first voxel + popcount), so no nVoxelsPSum walk and no voxel path either.  svo_build.hip (kMakeBricks) builds them.
//   FL = 0 (EMBED): child pointers carry the child's occupancy mask in bits 24-31 (ENABLE_EMBEDED_MASK,
//                   voxCommon.hpp:7-9); < 2^24 nodes; node offsets fit 32 bits.
//   FL = 1        : plain 32-bit child indices (up to 2^32-2 nodes, e.g. the 8192^3 non-DAG stress octree).  The
//                   reference reads a node's mask from the node when it is entered (voxCommon.hpp:353-356), i.e. two
//                   dependent misses per descent on an HBM-resident tree.  Here the 8 child masks sit in the parent's
//                   64-byte line next to the 8 child pointers (nVoxelsPSum moves to a cold array that only
//                   voxelIndexFromPath reads), so a descent is ONE line fetch, exactly like the embedded flavour; the
//                   node mask travels in a fifth stack dword.
//
// Same results as the reference's octreeTraverse_EfficientParametric
// (voxCommon.hpp:231-423): identical slab arithmetic, child order, tie-breaks and hit test.  What is
// different is everything the hardware cares about.  Measured on MI355X (DESIGN.md 5.3,
// profiles/r01_gfx950_issue_and_gather_costs.txt): on cache-resident DAG octrees the step is bound by instruction
// issue -- a VOP3 instruction costs the SIMD ~4.4 cycles, a branch ~10, and the ONE scalar ALU of a CU is shared by its
// four SIMDs (~3 cycles of a SIMD's time per SALU instruction once it saturates) -- and on an HBM-resident octree by
// the chip's random-line rate (50 G lines/s).  The kernel is built around few instructions per NODE VISIT, residency
// and lane utilisation:
//
//  * one loop iteration settles a whole node visit (all candidate children at once, see the step below) instead of one
//    candidate child as the reference's inner loop does, and pushes a node only when a later valid candidate exists;
//  * 16-byte stack entries.  The reference saves 32 bytes per level (voxCommon.hpp:202-212).  Here:
//      - slot index = tree level of the saved node.  Pending entries are ancestors of the current node,
//        hence at strictly increasing levels, so "which entries are pending" is a 32-bit mask in a
//        register; pop = highest set bit.  Neither sp nor the level is stored.
//      - scale = 2^-level is rebuilt from the level.
//      - the child a node is left through (3 bits) rides in the sign bits of tx1/ty1/tz1: a saved node was entered with
//        min(x1,y1,z1) >= 0, so its exit times are never negative (a -0.0 would come back as +0.0,
//        which no comparison or output can distinguish).
//      - nVoxelSkipped is not saved at all: the path of child indices (3 bits per level, one 64-bit
//        register = the hit voxel's morton code) is kept instead; the traversal reports that path and
//        the CONSUMER of the hit sums nVoxelsPSum along it (voxelIndexFromPath) in a dense kernel where
//        all 64 lanes walk together.  This also removes the nVoxelsPSum load from every descent.
//    An entry is {child reference (index | mask << 24), tx1, ty1, tz1} = one ds_write_b128.
//  * LDS ring per lane (slot = level & (slots - 1); 8 slots = 8 KiB per wave for the embedded flavour, 8 + mask words = 12 KiB for the tree flavour, 4 for plain
//    indices, see MVRT_RING_OF).
//    A push that lands on an occupied slot first evicts that (shallower) entry to an HBM spill array
//    laid out [level][lane] (coalesced 1 KiB rows); a pop of an evicted level reads it back.  Hot
//    pushes and pops near the leaves never leave LDS.
//  * Persistent waves with lane refill: a wave owns a cursor into the ray stream (grabbed in chunks with
//    one atomic per chunk); whenever at least REFILL_MIN lanes have finished it loads new rays into
//    exactly those lanes.  Long rays no longer hold 63 idle lanes hostage.  Per-ray results do not
//    depend on which lane or wave traced them.
#pragma once
#include "mvrt_common.h"

// LDS ring slots per lane: 8 for the embedded flavour (cache-resident DAG octrees: 5 to 7 resident waves per SIMD perform alike, fewer
// evictions to the HBM spill rows are worth +1.5 %), 4 for the flavours that walk HBM-resident octrees (every resident wave counts there, and
// they carry a second ring for the node masks)
#ifndef MVRT_RING_EMBED
#define MVRT_RING_EMBED 8
#endif
#ifndef MVRT_RING_TREE
#define MVRT_RING_TREE 8 // tree flavour (13 levels at 8192^3): 8 slots = 12 KiB per wave with the two mask words, 3 waves per SIMD; measured on config 5: 4 slots (any of 3-6
						 // waves per SIMD) 600 Mrays/s, 8 slots 630, 16 slots (1 wave per SIMD) 404 -- the evictions cost, the occupancy does not
#endif
#define MVRT_RING_OF( FL ) ( ( FL ) == 0 ? MVRT_RING_EMBED : ( ( FL ) == 2 ? MVRT_RING_TREE : 4 ) )
// refill once this many lanes are idle (or all of them).  Embedded flavour: 28 since a refill also replays the hint's path (r03: 20 / 28 / 36 -> 67.2 / 65.8 / 70.9 ms
// per 36 launches of the headline).  Tree flavour (config 5: every node visit waits for HBM, an idle lane is a lost fetch slot): 4 / 8 / 12 / 16 / 20 / 28 ->
// 679 / 706 / 706 / 694 / 676 / 640 Mrays/s: 12.  Plain indices: 20 as in round 2.
#ifndef MVRT_REFILL_MIN_EMBED
#define MVRT_REFILL_MIN_EMBED 28
#endif
#ifndef MVRT_REFILL_MIN_TREE
#define MVRT_REFILL_MIN_TREE 12
#endif
#ifndef MVRT_REFILL_MIN_OTHER
#define MVRT_REFILL_MIN_OTHER 20
#endif
#define MVRT_REFILL_MIN_OF( FL ) ( ( FL ) == 0 ? MVRT_REFILL_MIN_EMBED : ( ( FL ) == 2 ? MVRT_REFILL_MIN_TREE : MVRT_REFILL_MIN_OTHER ) )

// ---- start below the root (embedded flavour) -------------------------------------------------------------------------------------------
// The reference starts every ray at the root (voxCommon.hpp:306-312).  A secondary ray of the path tracer starts ON the voxel its path just
// hit, and for such a ray ~9 of the ~11 first node visits only walk back down to that voxel: at every node on the way the first candidate that is
// not behind the origin is the octant the origin lies in, and that octant is an ancestor of the hit voxel.  Which octant it is follows from
// the node's slab times alone -- bit a of the octant = tM_a < max( S, 0 ) (derivation at startBelowRoot) -- so, given the path of ANY voxel
// that exists (the HINT), the walk is replayed without touching the octree: the same fp32 operations per level as the node-visit step,
// compared against the hint's child index; it stops at the first level where the two differ (or after MVRT_HINT_MAX levels) and the normal
// traversal starts THERE, with the skipped ancestors on the stack.  An ancestor is stacked when a later candidate exists geometrically (the
// reference's own push rule, voxCommon.hpp:368,377-380: a superset of this kernel's "later VALID candidate" rule -- the extra entries pop
// into visits that find nothing, which changes no result); the node references of the ancestors come from prefix tables (one u32 reference
// per path prefix of 0..MVRT_HINT_MAX levels, stored behind the children array): independent gathers, no pointer chase.
// Results -- t, nMajor, the voxel path, and the descents count, to which the skipped levels are added -- are those of the walk from the root
// for ANY valid hint; a hint is only ever a prefix of the path of an existing voxel.
#define MVRT_NO_HINT 0xFFFFFFFFu
#ifndef MVRT_HINT_MAX
#define MVRT_HINT_MAX 7u // levels a hint carries = deepest prefix table (8^7 entries of 4 bytes); at most the ring's 8 slots
#endif
MVRT_HDI uint32_t hintLevelsOf( uint32_t levels ) { return levels == 0u ? 0u : ( levels - 1u < MVRT_HINT_MAX ? levels - 1u : MVRT_HINT_MAX ); }
MVRT_HDI uint32_t hintTabLevelsOf( uint32_t levels ) { return hintLevelsOf( levels ); }
MVRT_HDI uint32_t prefixTabOffset( uint32_t l ) { return 0x49249249u & ( ( 1u << ( 3u * l ) ) - 1u ); } // (8^l - 1) / 7: entries of the tables of fewer levels
MVRT_HDI uint64_t prefixTabEntries( uint32_t levels ) { return (uint64_t)prefixTabOffset( hintTabLevelsOf( levels ) + 1u ); }
// the hint of a ray that starts on the voxel at `path` (a full root -> voxel path, 3 bits per level)
// (masked to the hint's 3 bits per level: a code beyond the grid -- only a caller of mvrt_trace_batch_hinted can pass one -- must not index past the prefix tables)
MVRT_HDI uint32_t hintFromVoxelPath( uint64_t path, uint32_t levels )
{
	const uint32_t P = hintLevelsOf( levels );
	return (uint32_t)( path >> ( 3u * ( levels - P ) ) ) & ( ( 1u << ( 3u * P ) ) - 1u );
}

struct StreamHit
{
	float t;
	int nMajor;
	uint64_t path; // child indices root -> hit voxel, 3 bits per level (= the voxel's morton code); 0 on a miss
	uint32_t descents;
};

// vIndex of the voxel at `path` = sum of nVoxelsPSum along root -> voxel (voxCommon.hpp:388-391).  Done by the
// CONSUMER of a hit (dense kernels, every lane busy), not inside the divergent traversal loop.
//
// Octrees built by this library also carry the CELL INDEX (SvoDev::cellBlocks, kernels_setup.hip kFillCellIndex): nVoxelsPSum sums along a path are the RANK of
// the voxel in Morton order (the reference numbers voxels depth first, children in index order, voxKernel.cu:296-329), and the builder has that order in hand --
// so per occupied cell of the last-but-one level (a parent of voxels, per PATH, not per shared DAG node) { rank of its first voxel, mask of its voxels } is
// stored in a dense 512-entry array per occupied 8 x 8 x 8 block of cells, found through a dense table over the block codes: the index of a hit voxel is two
// gathers (the second one into lines that neighbouring hits share) + a popcount instead of a dependent 64-byte line per level below the top table.  Same
// integers as the walk for every voxel that exists; uploaded octrees (any nVoxelsPSum, no Morton order at hand) keep the walk.
MVRT_DI uint32_t voxelIndexFromPath( const SvoDev& s, uint64_t path )
{
	if( s.tree ) return (uint32_t)path; // tree flavour: the traversal already reports the voxel's index (first voxel of its parent + rank)
	if( s.cellBlocks )
	{
		const uint64_t cell = path >> 3;
		const uint32_t b = s.cellBlocks[cell >> s.cellBits];
		if( b != 0xFFFFFFFFu ) // (always, for the path of a voxel that exists)
		{
			const uint2 e = s.cellEntries[( (uint64_t)b << s.cellBits ) | ( (uint32_t)cell & ( ( 1u << s.cellBits ) - 1u ) )];
			return e.x + (uint32_t)__popc( e.y & ( ( 1u << ( (uint32_t)path & 7u ) ) - 1u ) );
		}
	}
	uint32_t n = s.rootIndex, v = 0, l0 = 0;
	if( s.topLevels ) // one table lookup replaces the first topLevels dependent gathers
	{
		const uint2 e = s.topTable[(uint32_t)( path >> ( 3u * ( s.levels - s.topLevels ) ) )];
		n = e.x;
		v = e.y;
		l0 = s.topLevels;
	}
	uint32_t ref = 0; // embedded flavour: the reference the walk arrived through (index | the node's own mask << 24), once one was read
	for( uint32_t l = l0; l < s.levels; l++ )
	{
		const uint32_t c = (uint32_t)( path >> ( 3u * ( s.levels - 1u - l ) ) ) & 7u;
		const Node64* nd = s.nodes + n;
		if( s.embedded )
		{
			if( l + 1u == s.levels && l > l0 && s.leafPsumIsPopcount )
			{
				// the parent of the voxel: its children are voxels, one each, so its nVoxelsPSum[c] is the number of its children before c --
				// a popcount of the mask that came with the reference to it: no fetch for the last level
				v += (uint32_t)__popc( ( ref >> 24 ) & ( ( 1u << c ) - 1u ) );
				break;
			}
			v += nd->psum[c];
			ref = nd->children[c];
			n = ref & 0xFFFFFFu;
		}
		else
		{
			v += s.psumCold[(uint64_t)n * 8 + c];
			n = nd->children[c];
		}
	}
	return v;
}

// What the traversal needs from the octree: kept small on purpose -- the fat SvoDev / PtParams structs cost
// ~100 SGPRs and made the compiler re-load the node pointer from kernarg memory inside the loop.
struct TraceCore
{
	const Node64* nodes;
	const uint32_t* kids; // embedded flavour: children[8] of every node, 32 bytes per node (the traversal never reads nVoxelsPSum: half the cache footprint)
	float lox, loy, loz, hix, hiy, hiz;
	uint32_t rootRef; // rootIndex | rootMask << 24 (voxCommon.hpp:306)
	uint32_t rootIndex, rootMask;
	uint32_t levelsM1; // tree flavour: levels - 1 = depth of the parents of voxels (they are "in-brick" nodes, like every second level above them)
};
MVRT_HDI TraceCore makeTraceCore( const SvoDev& s )
{
	TraceCore c;
	c.nodes = s.nodes;
	c.kids = s.kids;
	c.lox = s.lower.x; c.loy = s.lower.y; c.loz = s.lower.z;
	c.hix = s.upper.x; c.hiy = s.upper.y; c.hiz = s.upper.z;
	c.rootRef = s.rootIndex | ( s.rootMask << 24 );
	c.rootIndex = s.tree ? s.treeRoot : s.rootIndex;
	c.rootMask = s.rootMask;
	c.levelsM1 = s.levels - 1u;
	return c;
}

// Single-instruction helpers through inline asm: written as plain C the optimiser turns these mask selects back into
// v_cmp + v_cndmask pairs (two VALU slots and a VCC hazard each); the kernel is VALU-issue bound, so the forms matter.
MVRT_DI uint32_t bitMask( uint32_t v, uint32_t bit ) // 0 or 0xFFFFFFFF from bit `bit` of v (v_bfe_i32)
{
	uint32_t r;
	asm( "v_bfe_i32 %0, %1, %2, 1" : "=v"( r ) : "v"( v ), "v"( bit ) );
	return r;
}
MVRT_DI uint32_t lshlOr( uint32_t a, uint32_t sh, uint32_t c ) // (a << sh) | c  (v_lshl_or_b32)
{
	uint32_t r;
	asm( "v_lshl_or_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( sh ), "v"( c ) );
	return r;
}
MVRT_DI uint32_t andOr( uint32_t a, uint32_t m, uint32_t c ) // (a & m) | c  (v_and_or_b32)
{
	uint32_t r;
	asm( "v_and_or_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( a ), "v"( m ), "v"( c ) );
	return r;
}
typedef unsigned long long lmask;						   // one bit per lane, wave-uniform (an SGPR pair)
#define LANE( m ) __builtin_amdgcn_inverse_ballot_w64( m ) // this lane's bit of a lane mask, as a branch / select condition
// lane selects on a lane mask held in an SGPR pair, one v_cndmask_b32 each, issued as asm: written as nested ?: on LANE( m ) the
// compiler turns a chain of them into a chain of branches (~10 SIMD-cycles apiece).  The masks they read are produced by SALU
// instructions (no VALU-writes-SGPR hazard for the assembler-invisible reader).
MVRT_DI uint32_t selU( lmask m, uint32_t a, uint32_t b ) // lane bit set ? a : b
{
	uint32_t r;
	asm( "v_cndmask_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( b ), "v"( a ), "s"( m ) );
	return r;
}
#define MVRT_SELKK( m, K1, K0 ) ( { uint32_t r_; asm( "v_cndmask_b32 %0, " #K0 ", " #K1 ", %1" : "=v"( r_ ) : "s"( m ) ); r_; } )	   // bit ? K1 : K0
#define MVRT_SELK( m, K1, b ) ( { uint32_t r_; asm( "v_cndmask_b32 %0, %1, " #K1 ", %2" : "=v"( r_ ) : "v"( b ), "s"( m ) ); r_; } ) // bit ? K1 : b
// IEEE minimum of two NaN-free values as ONE v_min_f32 (fminf() makes the compiler quiet possible signalling NaNs first: a v_max x, x per operand)
MVRT_DI float minF( float a, float b )
{
	float r;
	asm( "v_min_f32 %0, %1, %2" : "=v"( r ) : "v"( a ), "v"( b ) );
	return r;
}
// lane mask of the lanes whose `v` has its sign bit set, one v_cmp at the point of use (as __ballot( (int)v < 0 ) on a value defined under a
// divergent branch the compiler builds the mask inside the branch and re-materialises it with a v_cndmask + v_cmp pair per use)
MVRT_DI unsigned long long signMask( uint32_t v )
{
	unsigned long long m;
	asm( "v_cmp_gt_i32 %0, 0, %1" : "=s"( m ) : "v"( v ) );
	return m;
}
typedef float v2f __attribute__( ( ext_vector_type( 2 ) ) );
typedef uint32_t u4v __attribute__( ( ext_vector_type( 4 ) ) );
typedef __attribute__( ( address_space( 3 ) ) ) u4v LdsU4; // (u4v: clang vector, assignable across address spaces)
typedef __attribute__( ( address_space( 3 ) ) ) uint32_t LdsU32;
MVRT_DI uint32_t bfi( uint32_t mask, uint32_t a, uint32_t b ) // (a & mask) | (b & ~mask)  (v_bfi_b32)
{
	uint32_t r;
	asm( "v_bfi_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( mask ), "v"( a ), "v"( b ) );
	return r;
}


// Exact-emulation path for IRREGULAR rays: a direction component of exactly zero makes the reference's slab
// arithmetic overflow (dt = +inf) and later produce inf - inf = NaN, which its compare-select max/min chains
// (vectorMath.hpp:100-108) propagate in an argument-order dependent way -- such rays typically "miss" in the
// reference even when they geometrically hit.  Parity means reproducing that, so rays whose slab deltas are not
// all finite never enter the fast loop (whose v_max3/v_min3 assume NaN-free data); they are traced here, one
// lane at a time if need be, with the reference's exact operation order and a stack in the HBM spill rows
// (two 16-byte rows per slot).  They are measure-zero in rendering; cost is irrelevant.
// Tree flavour, state of a node while it is walked:
//   brick root (odd distance from the voxels' parents): node = BASE = where the children of its first existing child start (bricks, or voxels at the
//     bottom), bLo / bHi = the masks of its eight children (byte c = child c, 0 = absent), nodeMask = its own mask (= the non-zero bytes).
//     The children's children lie back to back in child order (the builder numbers every level in Morton order), so child c's start is
//     BASE + popcount( masks of the children before c ): the 16 bytes { bLo, bHi, own mask, BASE } fetched when the brick is ENTERED are all a lane
//     ever needs of its line -- no second access for the descent into a child, and none when the walk comes back to the brick from its stack
//     (measured before: 98 line fetches per ray for 66 bricks entered, at 91 % of the chip's random-line rate);
//   in-brick node: node = where its children start, nodeMask = its mask.
MVRT_DI uint32_t nonZeroBytes( uint32_t lo, uint32_t hi ) // bit c = byte c of { lo, hi } is not zero
{
	const uint32_t a = ( ( ( lo & 0x7F7F7F7Fu ) + 0x7F7F7F7Fu ) | lo ) & 0x80808080u, b = ( ( ( hi & 0x7F7F7F7Fu ) + 0x7F7F7F7Fu ) | hi ) & 0x80808080u;
	const uint32_t la = ( a >> 7 ) | ( a >> 14 ) | ( a >> 21 ) | ( a >> 28 ), lb = ( b >> 7 ) | ( b >> 14 ) | ( b >> 21 ) | ( b >> 28 );
	return ( la & 15u ) | ( ( lb & 15u ) << 4 );
}
MVRT_DI void treeEnterBrick( const Node64* __restrict__ bricks, uint32_t at, uint32_t* node, uint32_t* nodeMask, uint32_t* bLo, uint32_t* bHi )
{
	const uint4 q = ( (const uint4*)bricks )[at]; // THE dependent fetch of two levels: a 16-byte brick (`nodes` of a tree-flavour octree are bricks, not Node64 lines)
	*bLo = q.x;
	*bHi = q.y;
	*nodeMask = q.z;
	*node = q.w;
}
// one descent: from an in-brick node onto the child's brick -- or onto a voxel, whose index is recorded -- and from a brick root to one of its children
MVRT_DI void treeDescend( const Node64* __restrict__ bricks, uint32_t levelsM1, uint32_t level, uint32_t childIndex, uint32_t* node, uint32_t* nodeMask, uint32_t* bLo, uint32_t* bHi,
						  uint64_t* leafV )
{
	if( ( ( levelsM1 - level ) & 1u ) == 0u ) // in-brick node
	{
		const uint32_t at = *node + (uint32_t)__popc( *nodeMask & ( ( 1u << childIndex ) - 1u ) );
		if( level == levelsM1 ) // its children are voxels
		{
			*leafV = at;
			*node = MVRT_LEAF;
		}
		else
			treeEnterBrick( bricks, at, node, nodeMask, bLo, bHi );
	}
	else // brick root -> child: arithmetic only
	{
		const uint32_t sh = 8u * ( childIndex & 3u );
		const uint32_t word = childIndex < 4u ? *bLo : *bHi;
		const uint32_t below = ( childIndex < 4u ? 0u : (uint32_t)__popc( *bLo ) ) + (uint32_t)__popc( word & ( ( 1u << sh ) - 1u ) );
		*nodeMask = ( word >> sh ) & 0xFFu;
		*node = *node + below;
	}
}

template <int FL>
MVRT_DI void traceIrregular( const TraceCore& s, float tx1, float ty1, float tz1, float t0x, float t0y, float t0z, uint32_t vMask, uint4* __restrict__ mySpill,
							 uint64_t spillStride, float* resT, int* resN, uint64_t* pathOut, uint32_t* descentsOut )
{
	constexpr bool EMBED = FL == 0, TREE = FL == 2;
	const float dtx = tx1 - t0x, dty = ty1 - t0y, dtz = tz1 - t0z;
	uint32_t node = EMBED ? s.rootRef : s.rootIndex, nodeMask = s.rootMask, level = 0, childMask = 8u, sp = 0, descents = 0, bLo = 0, bHi = 0;
	uint64_t path = 0;
	if( TREE && ( s.levelsM1 & 1u ) ) treeEnterBrick( s.nodes, s.rootIndex, &node, &nodeMask, &bLo, &bHi ); // the root is a brick root
	for( ;; )
	{
		const float scale = mvrt_u2f( ( 127u - level ) << 23 );
		const float tx0 = tx1 - dtx * scale;
		const float ty0 = ty1 - dty * scale;
		const float tz0 = tz1 - dtz * scale;
		const float S = max3f( tx0, ty0, tz0 );
		bool pop = false;
		if( node == MVRT_LEAF )
		{
			if( 0.0f < S )
			{
				*resT = S;
				*resN = ( S == tx0 ) ? 1 : ( ( S == ty0 ) ? 2 : 0 );
				break;
			}
			pop = true;
		}
		else
		{
			const float txM = 0.5f * ( tx0 + tx1 );
			const float tyM = 0.5f * ( ty0 + ty1 );
			const float tzM = 0.5f * ( tz0 + tz1 );
			if( childMask & 8u ) childMask = ( txM < S ? 1u : 0u ) | ( tyM < S ? 2u : 0u ) | ( tzM < S ? 4u : 0u );
			const float x1 = ( childMask & 1u ) ? tx1 : txM;
			const float y1 = ( childMask & 2u ) ? ty1 : tyM;
			const float z1 = ( childMask & 4u ) ? tz1 : tzM;
			const float u = min3f( x1, y1, z1 );
			const uint32_t mv = ( u == x1 ) ? 1u : ( ( u == y1 ) ? 2u : 4u );
			const bool hasNext = ( childMask & mv ) == 0;
			const uint32_t childIndex = childMask ^ vMask;
			const uint32_t nextMask = childMask | mv;
			const bool exists = EMBED ? ( ( node >> ( 24u + childIndex ) ) & 1u ) != 0 : ( ( nodeMask >> childIndex ) & 1u ) != 0;
			if( exists && !( u < 0.0f ) )
			{
				if( hasNext )
				{
					uint4 a, b;
					a.x = node;
					a.y = mvrt_f2u( tx1 );
					a.z = mvrt_f2u( ty1 );
					a.w = mvrt_f2u( tz1 );
					b.x = nextMask | ( level << 3 );
					b.y = TREE ? bLo : (uint32_t)path; // (the tree flavour records no path: the voxel's index comes with the last descent)
					b.z = TREE ? bHi : (uint32_t)( path >> 32 );
					b.w = nodeMask;
					mySpill[(uint64_t)( 2 * sp ) * spillStride] = a;
					mySpill[(uint64_t)( 2 * sp + 1 ) * spillStride] = b;
					sp++;
				}
				if( EMBED )
				{
					node = s.kids[( node & 0xFFFFFFu ) * 8u + childIndex];
				}
				else if( TREE )
				{
					treeDescend( s.nodes, s.levelsM1, level, childIndex, &node, &nodeMask, &bLo, &bHi, &path );
				}
				else
				{
					const Node64* nd = s.nodes + node;
					nodeMask = ( nd->psum[childIndex >> 2] >> ( 8u * ( childIndex & 3u ) ) ) & 0xFFu; // the child's mask, same line
					node = nd->children[childIndex];
				}
				descents++;
				if( !TREE ) path = ( path << 3 ) | childIndex;
				tx1 = x1;
				ty1 = y1;
				tz1 = z1;
				level++;
				childMask = 8u;
			}
			else if( hasNext )
			{
				childMask = nextMask;
			}
			else
			{
				pop = true;
			}
		}
		if( pop )
		{
			if( sp == 0 ) break;
			sp--;
			const uint4 a = mySpill[(uint64_t)( 2 * sp ) * spillStride];
			const uint4 b = mySpill[(uint64_t)( 2 * sp + 1 ) * spillStride];
			node = a.x;
			tx1 = mvrt_u2f( a.y );
			ty1 = mvrt_u2f( a.z );
			tz1 = mvrt_u2f( a.w );
			childMask = b.x & 7u;
			level = ( b.x >> 3 ) & 31u;
			if( TREE )
			{
				bLo = b.y;
				bHi = b.z;
			}
			else
				path = (uint64_t)b.y | ( (uint64_t)b.z << 32 );
			nodeMask = b.w;
		}
	}
	*pathOut = path;
	*descentsOut = descents;
}

// IO concept (ray indices are 32-bit: a launch never exceeds 2^32 rays):
//   bool load( uint32_t ray, f3* ro, f3* rd, uint32_t* hint )   -> returns isShadowRay; *hint = MVRT_NO_HINT or hintFromVoxelPath( an existing voxel )
//   void store( uint32_t ray, const StreamHit& h, bool isShadowRay )
//
// One loop iteration = (1) refill when enough lanes are idle: first STORE the results those lanes still hold
// (one store site, many lanes per store instruction), then load new rays into them; (2) one traversal step for
// every active lane: straight-line bit arithmetic (v_bfi / v_bfe selects instead of compare-select chains)
// followed by three shallow branches: descend (with push), pop, hit.
template <int FL, class IO>
MVRT_DI void traceStream( const TraceCore& s, IO& io, uint64_t total64, unsigned long long* __restrict__ cursor, uint32_t chunk, uint4* __restrict__ ldsRing /* [MVRT_RING_OF( FL )][64] */,
						  uint4* __restrict__ spill /* [levels][spillStride] */, uint64_t spillStride, uint64_t spillLane, uint32_t* __restrict__ ldsMask = nullptr /* [MVRT_RING][64], !EMBED */,
						  uint32_t* __restrict__ spillMask = nullptr /* [levels][spillStride], !EMBED */, uint32_t* __restrict__ spillMask2 = nullptr /* second word, TREE */ )
{
	constexpr bool EMBED = FL == 0, TREE = FL == 2;
	constexpr uint32_t MVRT_RING = MVRT_RING_OF( FL );
	constexpr uint32_t MVRT_RING_CLASH = MVRT_RING == 4 ? 0x11111111u : ( MVRT_RING == 8 ? 0x01010101u : 0x00010001u ); // the levels that share ring slot 0
	const uint32_t lane = threadIdx.x;
	const uint32_t total = (uint32_t)total64;
	const Node64* __restrict__ nodes = s.nodes;
	const uint32_t* __restrict__ kids = s.kids;
	// LDS ring: explicit LDS address space (so the optimiser cannot fold a ring read and a spill read into one flat
	// load); slot k of this lane at byte offset k * 1024 + lane * 16
	LdsU4* const myRing = (LdsU4*)ldsRing + lane;
	const uint32_t ringAddr = (uint32_t)(uintptr_t)myRing; // LDS byte address of this lane's slot 0
	uint4* const mySpill = spill + spillLane; // level L at mySpill[L * spillStride]  (irregular rays only)
	LdsU32* const myRingMask = EMBED ? nullptr : (LdsU32*)ldsMask + lane;
	// tree flavour: a stacked brick root keeps its 8 child masks in the two mask words (its own mask = their non-zero bytes); an in-brick node its mask in the first
	LdsU32* const myRingMask2 = TREE ? (LdsU32*)ldsMask + MVRT_RING * 64 + lane : nullptr;
	// spill rows: spillStride is a power of two (traceWorkspaceLanes), rows * stride * 16 B < 4 GiB: row L of this lane
	// is base + ((L << spillShift) + lane offset) with 32-bit arithmetic and a scalar base
	const uint32_t spillShift = 4u + (uint32_t)__builtin_ctzll( spillStride );
	const uint32_t spillOff = (uint32_t)spillLane * 16u;
	const uint32_t spillMaskShift = spillShift - 2u, spillMaskOff = (uint32_t)spillLane * 4u;

	// wave-uniform cursor state
	uint32_t chunkNext = 0, chunkEnd = 0;
	bool exhausted = false;

	// per-lane state.  st: 0 idle, 1 traversing, finished and holding a result that is not stored yet: 2 = miss, 3 = hit (the lane's slab state is
	// untouched since the step that found the leaf: t and nMajor are re-derived from it at flush time instead of being selected into two more
	// registers in EVERY step), 4 = irregular ray (result parked in tx1 / level)
	uint32_t st = 0;
	bool isShadow = false;
	uint32_t ray = 0;
	float dtx = 0, dty = 0, dtz = 0, tx1 = 0, ty1 = 0, tz1 = 0;
	uint32_t vMask = 0, vMaskHi = 24u, node = 0, nodeMask = 0, level = 0, childMask = 8u, pending = 0, inLds = 0, descents = 0;
	uint32_t bLo = 0, bHi = 0; // tree flavour: child masks of the brick whose root is being visited
	uint64_t path = 0;
	// result of a finished lane (st >= 2), from its parked state
	auto finishedHit = [&]( StreamHit* h ) {
		h->t = MVRT_MAXF;
		h->nMajor = -1;
		if( st == 3u ) // :324-334 -- the same arithmetic as the step that accepted the leaf
		{
			const float scale = mvrt_u2f( ( 127u - level ) << 23 );
			const float tx0 = tx1 - dtx * scale, ty0 = ty1 - dty * scale, tz0 = tz1 - dtz * scale;
			const float S = fmaxf( fmaxf( tx0, ty0 ), tz0 );
			h->t = S;
			h->nMajor = ( S == tx0 ) ? 1 : ( ( S == ty0 ) ? 2 : 0 );
		}
		else if( st == 4u )
		{
			h->t = tx1;
			h->nMajor = (int)level;
		}
		h->path = h->t != MVRT_MAXF ? path : 0ull;
		h->descents = descents;
	};

#ifdef MVRT_UTIL_STATS
	uint32_t utilVis = 0; // node-visit iterations of the lane's current ray
	const unsigned long long utilT0 = clock64();
#endif
	for( ;; )
	{
#ifdef MVRT_UTIL_STATS
		const unsigned long long utilTr = clock64(); // shader clocks this wave spends in refill sections (flush of results + ray loads + setup + hint replay)
#endif
		// ---------------- (1) refill: control only gets here when enough lanes are idle ----------------
		const unsigned long long idleMask = __ballot( st != 1u );
		const uint32_t nIdle = __popcll( idleMask );
		{
			if( st >= 2u ) // flush results of the lanes that finished since the last refill
			{
				StreamHit h;
				finishedHit( &h );
				io.store( ray, h, isShadow );
				st = 0;
			}
			if( !exhausted )
			{
				uint32_t need = nIdle;
				const uint32_t myRank = __popcll( idleMask & ( ( 1ull << lane ) - 1ull ) );
				uint32_t given = 0;
				while( need > 0 )
				{
					if( chunkNext == chunkEnd )
					{
						unsigned long long base = 0;
						const uint32_t c = chunk;
						if( lane == 0 ) base = atomicAdd( cursor, (unsigned long long)c );
						// readfirstlane, not a shuffle: the cursor state (and with it `exhausted` and the exit test of the step loop) is then
						// wave-uniform FOR THE COMPILER -- scalar registers, scalar branches, and the lane masks of the step loop stay in SGPRs
						// across its exit instead of being copied to VGPRs in every iteration
						base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane( (uint32_t)base ) | ( (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane( (uint32_t)( base >> 32 ) ) << 32 );
						if( base >= total )
						{
							exhausted = true;
							// a wave that only drains its last rays is on the launch's critical path: let it win issue arbitration against the waves of
							// a sibling pass that shares the SIMD (1/8 tile share: -1 % dragon, -3 % rtcamp; nothing on a full frame)
							__builtin_amdgcn_s_setprio( 3 );
							break;
						}
						chunkNext = (uint32_t)base;
						chunkEnd = ( total - chunkNext ) < c ? total : chunkNext + c;
					}
					const uint32_t avail = chunkEnd - chunkNext;
					const uint32_t take = avail < need ? avail : need;
					if( st == 0u && myRank >= given && myRank < given + take )
					{
						ray = chunkNext + ( myRank - given );
						// ---- ray setup, voxCommon.hpp:240-312 ----
						f3 ro, rd;
						uint32_t hint = MVRT_NO_HINT;
						isShadow = io.load( ray, &ro, &rd, &hint );
						float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;
						vMask = 0;
						if( ix < 0.0f )
						{
							vMask |= 1u;
							ix = -ix;
							ro.x = s.lox + s.hix - ro.x;
						}
						if( iy < 0.0f )
						{
							vMask |= 2u;
							iy = -iy;
							ro.y = s.loy + s.hiy - ro.y;
						}
						if( iz < 0.0f )
						{
							vMask |= 4u;
							iz = -iz;
							ro.z = s.loz + s.hiz - ro.z;
						}
						ix = smin( ix, MVRT_MAXF / smax( smax( sabs( s.lox - ro.x ), sabs( s.hix - ro.x ) ), 1.0f ) );
						iy = smin( iy, MVRT_MAXF / smax( smax( sabs( s.loy - ro.y ), sabs( s.hiy - ro.y ) ), 1.0f ) );
						iz = smin( iz, MVRT_MAXF / smax( smax( sabs( s.loz - ro.z ), sabs( s.hiz - ro.z ) ), 1.0f ) );
						const float t0x = ( s.lox - ro.x ) * ix, t0y = ( s.loy - ro.y ) * iy, t0z = ( s.loz - ro.z ) * iz;
						tx1 = ( s.hix - ro.x ) * ix;
						ty1 = ( s.hiy - ro.y ) * iy;
						tz1 = ( s.hiz - ro.z ) * iz;
						descents = 0;
						path = 0;
#ifdef MVRT_UTIL_STATS
						io.utilMaxRayIters = utilVis > io.utilMaxRayIters ? utilVis : io.utilMaxRayIters;
						utilVis = 0;
#endif
						vMaskHi = vMask | 24u;
						if( min3f( tx1, ty1, tz1 ) < max3f( t0x, t0y, t0z ) ) // :275-278 misses the root box
						{
							st = 2u;
						}
						else if( ( ( mvrt_f2u( tx1 - t0x ) & 0x7F800000u ) == 0x7F800000u ) || ( ( mvrt_f2u( ty1 - t0y ) & 0x7F800000u ) == 0x7F800000u ) ||
								 ( ( mvrt_f2u( tz1 - t0z ) & 0x7F800000u ) == 0x7F800000u ) )
						{
							// irregular ray (inf / NaN slab delta): exact reference emulation, see traceIrregular
							float resT = MVRT_MAXF;
							int resN = -1;
							traceIrregular<FL>( s, tx1, ty1, tz1, t0x, t0y, t0z, vMask, mySpill, spillStride, &resT, &resN, &path, &descents );
							tx1 = resT; // parked where finishedHit() looks for an st == 4 result
							level = (uint32_t)resN;
							st = 4u;
						}
						else
						{
							dtx = tx1 - t0x;
							dty = ty1 - t0y;
							dtz = tz1 - t0z;
							node = EMBED ? s.rootRef : s.rootIndex;
							nodeMask = s.rootMask;
							if( TREE && ( s.levelsM1 & 1u ) ) treeEnterBrick( nodes, s.rootIndex, &node, &nodeMask, &bLo, &bHi ); // the root is a brick root
							level = 0;
							childMask = 8u;
							pending = 0;
							inLds = 0;
							st = 1u;
							if( EMBED && hint != MVRT_NO_HINT && !( min3f( tx1, ty1, tz1 ) < 0.0f ) ) // (a box behind the origin: the root visit settles it)
							{
								// Start below the root: replay the walk down the hint's path (see the top of this file).  In a first visit with
								// T = min( t1 ) >= 0 the candidates behind the origin are those ended by a NEGATIVE mid-plane event (kx, ky, kz of the
								// step below; the "before the first exit" condition of a flip holds for every negative tM once T >= 0), and they are
								// a prefix of the candidate order; so the first candidate that is not behind has bit a = ( tM_a < S ) | ( tM_a < 0 )
								// = tM_a < max( S, 0 ).  If that octant is the hint's child it exists, the step would enter it (its exit is >= 0 by the
								// same inequalities: T stays >= 0 level after level), and a later candidate exists iff an axis whose bit is clear flips
								// before the first exit event (fX, fY, fZ of the step).
								// Straight-line code: the level counter is a compile-time constant (the loop is unrolled, at most MVRT_HINT_MAX = 7
								// levels = ring slots: a refilled lane's ring is empty, so nothing is ever evicted here), the node references of the
								// hint's ancestors are ONE batch of independent table gathers issued up front (no pointer chase), lanes that have left
								// the hint's path just stop changing their state.
								const uint32_t P = hintLevelsOf( s.levelsM1 + 1u );
								const char* const tab = (const char*)kids + ( ( s.rootIndex + 1u ) << 5 ); // the prefix tables lie behind the children array (32 B per node; root = last node)
								uint32_t anc[MVRT_HINT_MAX + 1];
#pragma unroll
								for( uint32_t L = 1; L <= MVRT_HINT_MAX; L++ )
									anc[L] = L <= P ? *(const uint32_t*)( tab + ( ( prefixTabOffset( L ) + ( hint >> ( 3u * ( P - L ) ) ) ) << 2 ) ) : 0u;
								uint32_t k = 0u;
								bool on = true;
								anc[0] = node;
#pragma unroll
								for( uint32_t L = 0; L < MVRT_HINT_MAX; L++ )
								{
									if( L < P && on ) // (L < P is wave-uniform; lanes that have left the hint's path drop out of the rest of the nest)
									{
										const float scale = mvrt_u2f( ( 127u - L ) << 23 );
										const v2f t1yz = { ty1, tz1 };
										const v2f dtyz = { dty, dtz };
										const float tx0 = tx1 - dtx * scale; // :317-320, the step's own operations
										const v2f t0yz = t1yz - dtyz * scale;
										const float S0 = fmaxf( fmaxf( fmaxf( tx0, t0yz.x ), t0yz.y ), 0.0f );
										const float txM = 0.5f * ( tx0 + tx1 ); // :338-340
										const v2f tMyz = ( t0yz + t1yz ) * 0.5f;
										const bool bx = txM < S0, by = tMyz.x < S0, bz = tMyz.y < S0;
										const uint32_t b = ( bx ? 1u : 0u ) | ( by ? 2u : 0u ) | ( bz ? 4u : 0u );
										on = ( b ^ vMask ) == ( ( hint >> ( 3u * ( P - 1u - L ) ) ) & 7u ); // the origin's octant is the hint's child: it exists, the step enters it
										if( on )
										{
											const bool later = ( (int)!bx & (int)( txM <= minF( ty1, tz1 ) ) ) | ( (int)!by & (int)( tMyz.x < tx1 ) & (int)( tMyz.x <= tz1 ) ) | ( (int)!bz & (int)( tMyz.y < minF( tx1, ty1 ) ) );
											if( later ) // stack the ancestor (:377-380); its exit times carry the octant it is left through, like any entry
											{
												// (the node reference -- word 0 -- is filled in after the walk, when the table gathers have landed: their
												// latency hides behind this arithmetic instead of stalling the wave at the first level)
												LdsU32* const w = (LdsU32*)( myRing + L * 64 );
												w[1] = mvrt_f2u( tx1 ) | ( bx ? 0x80000000u : 0u ); // (T >= 0: the sign bits are free)
												w[2] = mvrt_f2u( ty1 ) | ( by ? 0x80000000u : 0u );
												w[3] = mvrt_f2u( tz1 ) | ( bz ? 0x80000000u : 0u );
												pending |= 1u << L;
											}
											tx1 = bx ? tx1 : txM; // :382-386
											ty1 = by ? ty1 : tMyz.x;
											tz1 = bz ? tz1 : tMyz.y;
											k = L + 1u;
										}
									}
								}
#pragma unroll
								for( uint32_t L = 0; L < MVRT_HINT_MAX; L++ )
								{
									if( L < P )
									{
										if( pending & ( 1u << L ) ) *(LdsU32*)( myRing + L * 64 ) = anc[L];
										node = k == L + 1u ? anc[L + 1] : node;
									}
								}
								inLds = pending;
								level = k;
								path = k ? (uint64_t)( hint >> ( 3u * ( P - k ) ) ) : 0ull;
								descents = k; // the reference fetched one child pointer per level (:381)
							}
						}
					}
					given += take;
					need -= take;
					chunkNext += take;
				}
			}
			if( __ballot( st == 1u ) == 0ull )
			{
				if( !exhausted ) continue; // every ray just loaded missed the root box: go round again
				if( st >= 2u )			   // final flush
				{
					StreamHit h;
					finishedHit( &h );
					io.store( ray, h, isShadow );
				}
#ifdef MVRT_UTIL_STATS
				io.utilMaxRayIters = utilVis > io.utilMaxRayIters ? utilVis : io.utilMaxRayIters;
				io.utilRefillClocks += clock64() - utilTr;
				io.utilTotalClocks += clock64() - utilT0;
#endif
				break; // every lane idle and the stream is empty: the wave retires
			}
		}

#ifdef MVRT_UTIL_STATS
		io.utilRefillClocks += clock64() - utilTr;
		{
			const unsigned long long am = __ballot( st == 1u );
			if( lane == 0 )
			{
				io.utilIters++;
				io.utilActive += __popcll( am );
				io.utilTailIters += exhausted ? 1u : 0u;
				io.utilTailActive += exhausted ? __popcll( am ) : 0u;
			}
		}
#endif
		// ---------------- (2) traversal steps until enough lanes are idle again ----------------
		// NODE-VISIT step.  The reference examines the candidate children of a node one at a time (voxCommon.hpp:362-412); measured on
		// a path-traced bunny that is 30.3 candidate tests per ray for 20.6 node visits and 14.7 descents, and 2.0 of the 5.5 pops lead
		// to no further descent.  Here ONE step settles a whole visit: the order in which the ray leaves the octants of a node is a
		// pure function of the three mid-plane times and the three exit times -- walk the events (time, axis) in lexicographic order
		// (the reference's min + "x first, then y" tie rule) until the first exit event; every earlier mid-plane event of a not yet
		// passed axis is a flip -- so the up to four candidates, their existence in the node's mask and the "behind the origin" test
		// are evaluated side by side, the first valid one is entered, and the node is pushed only if a later VALID candidate exists.
		// All of it is 1-bit logic: the compares produce 64-bit lane masks and the combinatorics run on the scalar unit.  The child
		// mask of every lane lives bit-sliced in three SGPR pairs (+ a "first visit" mask) while the loop runs.
		lmask cmX = __ballot( ( childMask & 1u ) != 0u ), cmY = __ballot( ( childMask & 2u ) != 0u ), cmZ = __ballot( ( childMask & 4u ) != 0u );
		lmask mFirst = __ballot( ( childMask & 8u ) != 0u );
		lmask actM = __ballot( st == 1u ), hitM = 0ull, missM = 0ull;
		for( ;; )
		{
			const lmask act = actM;
#ifdef MVRT_UTIL_STATS
			utilVis += LANE( act ) ? 1u : 0u;
			if( lane == 0 ) // per wave-ITERATION tallies (the block above counts refill events)
			{
				io.utilTailIters++;
				io.utilTailActive += (unsigned long long)__popcll( act );
			}
#endif
			const float scale = mvrt_u2f( ( 127u - level ) << 23 );
			const v2f t1yz = { ty1, tz1 };
			const v2f dtyz = { dty, dtz };
			const float tx0 = tx1 - dtx * scale; // :317-320
			const v2f t0yz = t1yz - dtyz * scale;
			const float ty0 = t0yz.x, tz0 = t0yz.y;
			const float S = fmaxf( fmaxf( tx0, ty0 ), tz0 );
			const float txM = 0.5f * ( tx0 + tx1 ); // :338-340
			const v2f tMyz = ( t0yz + t1yz ) * 0.5f;
			const float tyM = tMyz.x, tzM = tMyz.y;
			// octant the node is entered in (:342-348), or the candidate a popped node resumes with
			const lmask X = ( mFirst & __ballot( txM < S ) ) | ( ~mFirst & cmX );
			const lmask Y = ( mFirst & __ballot( tyM < S ) ) | ( ~mFirst & cmY );
			const lmask Z = ( mFirst & __ballot( tzM < S ) ) | ( ~mFirst & cmZ );
			// first exit event = lexicographic minimum of (t1, axis); flips = mid-plane events of unset axes that come before it, i.e.
			// before EVERY exit event (an axis' own exit never precedes its mid-plane: tM <= t1).  (tM_a, a) < (t1_b, b) is "tM_a <= t1_b" for
			// a < b and "tM_a < t1_b" for a > b
			const float mXY = minF( tx1, ty1 ), mYZ = minF( ty1, tz1 );
			const float T = minF( mXY, tz1 );
			const lmask fX = ~X & __ballot( txM <= mYZ );
			const lmask fY = ~Y & __ballot( tyM < tx1 ) & __ballot( tyM <= tz1 );
			const lmask fZ = ~Z & __ballot( tzM < mXY );
			// order of the flips among themselves: (tM, axis) lexicographic
			const lmask xy = __ballot( txM <= tyM ), xz = __ballot( txM <= tzM ), yz = __ballot( tyM <= tzM );
			const lmask n1 = fX | fY | fZ, n2 = ( fX & fY ) | ( fZ & ( fX | fY ) ), n3 = fX & fY & fZ; // a 2nd / 3rd / 4th candidate exists geometrically
			// the axis of the FIRST flip, and -- only when all three axes flip -- of the LAST one
			const lmask aX = fX & ~( fY & ~xy ) & ~( fZ & ~xz );
			const lmask aY = fY & ~( fX & xy ) & ~( fZ & ~yz );
			const lmask aZ = fZ & ~( fX & xz ) & ~( fY & yz );
			const lmask zX = n3 & ~( xy | xz ), zY = n3 & xy & ~yz, zZ = n3 & xz & yz;
			// child indices of the four candidates (mirrored space), nested: each adds the axis of one flip.  With two flips the third
			// candidate is already the last (i2 == i3); with three, it is the last minus the last flip
			// (asm selects = one v_cndmask on an SGPR mask; written as nested ?: the compiler turns these chains into branches)
			const uint32_t i0 = MVRT_SELKK( X, 1, 0 ) | MVRT_SELKK( Y, 2, 0 ) | MVRT_SELKK( Z, 4, 0 );
			const uint32_t i3 = i0 | MVRT_SELKK( fX, 1, 0 ) | MVRT_SELKK( fY, 2, 0 ) | MVRT_SELKK( fZ, 4, 0 );
			const uint32_t i1 = i0 | MVRT_SELK( aX, 1, MVRT_SELK( aY, 2, MVRT_SELKK( aZ, 4, 0 ) ) );
			const uint32_t i2 = i3 ^ MVRT_SELK( zX, 1, MVRT_SELK( zY, 2, MVRT_SELKK( zZ, 4, 0 ) ) );
#define MVRT_EXISTS( i ) __ballot( EMBED ? bitMask( node, ( i ) ^ vMaskHi ) != 0u : ( ( nodeMask >> ( ( ( i ) ^ vMaskHi ) & 7u ) ) & 1u ) != 0u )
			const lmask e0 = MVRT_EXISTS( i0 ), e1 = MVRT_EXISTS( i1 ), e2 = MVRT_EXISTS( i2 ), e3 = MVRT_EXISTS( i3 );
#undef MVRT_EXISTS
			// a candidate is behind the origin when the event that ends it is negative (:373).  Events are visited in time order, so the
			// candidates behind the origin are a prefix: candidate k is behind iff more than k flips are negative, all of them iff the
			// exit is
			const lmask kx = fX & __ballot( txM < 0.0f ), ky = fY & __ballot( tyM < 0.0f ), kz = fZ & __ballot( tzM < 0.0f );
			const lmask b0 = kx | ky | kz, b1 = ( kx & ky ) | ( ( kx | ky ) & kz ), b2 = kx & ky & kz;
			const lmask mLeaf = act & __ballot( node == MVRT_LEAF ); // :322
			const lmask inner = act & ~mLeaf & ~__ballot( T < 0.0f );
			// candidate 0 of a popped node is the child it was left through: already done
			const lmask v0 = mFirst & e0 & ~b0, v1 = n1 & e1 & ~b1, v2 = n2 & e2 & ~b2, v3 = n3 & e3;
			const lmask mGo = inner & ( v0 | v1 | v2 | v3 );
			const lmask l3 = v3, l2 = v2 | l3, l1 = v1 | l2; // a valid candidate at or after 3 / 2 / 1
			const lmask mPush = inner & ( ( v0 & l1 ) | ( ~v0 & ( ( v1 & l2 ) | ( ~v1 & v2 & l3 ) ) ) ); // only if a later VALID candidate exists (the reference: any later candidate; measured +8.6 %)
			const lmask mHit = mLeaf & __ballot( 0.0f < S ); // :324
			const lmask mPop = act & ~mHit & ~mGo;
			// the entered candidate = the first valid one
			const uint32_t ci = selU( v0, i0, selU( v1, i1, selU( v2, i2, i3 ) ) );
			const uint32_t childBit = ci ^ vMaskHi; // :369 (+24)
			const uint32_t childIndex = childBit & 7u;

			uint4 popped = make_uint4( 0u, 0u, 0u, 0u );
			uint32_t poppedMask = 0;
			const lmask mPopOk = mPop & __ballot( pending != 0u ); // (lane masks are only ever computed at the top level: a value assigned
																  // under a divergent branch stops being wave-uniform for the compiler)
			// (pop before descend: the two touch disjoint lanes, and the child-pointer load issued by the descent is then the last
			// thing of the iteration -- its latency overlaps with the slab arithmetic of the next one instead of being waited for here)
			if( LANE( mPopOk ) ) // :414-422 (a lane that has to pop with an empty stack has missed: handled with the hits below)
			{
				{
					const uint32_t L = 31u - __builtin_clz( pending );
					const uint32_t bit = 1u << L;
					u4v ev;
					asm volatile( "ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"( ev ) : "v"( ringAddr + ( ( L & ( MVRT_RING - 1 ) ) << 10 ) ) : "memory" );
					popped = make_uint4( ev.x, ev.y, ev.z, ev.w );
					uint32_t poppedMask2 = 0;
					if( !EMBED ) poppedMask = myRingMask[( L & ( MVRT_RING - 1 ) ) * 64];
					if( TREE ) poppedMask2 = myRingMask2[( L & ( MVRT_RING - 1 ) ) * 64];
					if( !( inLds & bit ) ) // rare; the empty asm keeps this a real branch (otherwise: address select + one flat load)
					{
						asm volatile( "" ::: "memory" );
						popped = *(const uint4*)( (const char*)spill + ( ( L << spillShift ) + spillOff ) );
						if( !EMBED ) poppedMask = *(const uint32_t*)( (const char*)spillMask + ( ( L << spillMaskShift ) + spillMaskOff ) );
						if( TREE ) poppedMask2 = *(const uint32_t*)( (const char*)spillMask2 + ( ( L << spillMaskShift ) + spillMaskOff ) );
					}
					if( TREE && ( ( s.levelsM1 - L ) & 1u ) ) // back at a brick root: its child masks come off the stack, its own mask is their non-zero bytes
					{
						bLo = poppedMask;
						bHi = poppedMask2;
						poppedMask = nonZeroBytes( bLo, bHi );
					}
					pending &= ~bit;
					inLds &= ~bit;
					if( !TREE ) path >>= 3u * ( level - L );
					level = L;
					node = popped.x;

					if( !EMBED ) nodeMask = poppedMask;
					tx1 = mvrt_u2f( popped.y & 0x7FFFFFFFu );
					ty1 = mvrt_u2f( popped.z & 0x7FFFFFFFu );
					tz1 = mvrt_u2f( popped.w & 0x7FFFFFFFu );
				}
			}
			if( LANE( mGo ) )
			{
				if( LANE( mPush ) ) // push (:377-380)
				{
					const uint32_t slot = level & ( MVRT_RING - 1 );
					const uint32_t clash = inLds & ( MVRT_RING_CLASH << slot );
					if( clash ) // the slot still holds a shallower pending entry: evict it to HBM
					{
						const uint32_t lc = __builtin_ctz( clash );
						*(u4v*)( (char*)spill + ( ( lc << spillShift ) + spillOff ) ) = myRing[slot * 64];
						if( !EMBED ) *(uint32_t*)( (char*)spillMask + ( ( lc << spillMaskShift ) + spillMaskOff ) ) = myRingMask[slot * 64];
						if( TREE ) *(uint32_t*)( (char*)spillMask2 + ( ( lc << spillMaskShift ) + spillMaskOff ) ) = myRingMask2[slot * 64];
						inLds &= ~clash;
					}
					u4v e;
					e.x = node;
					// the sign bits of a saved node's exit times are free (entered with min >= 0): they carry the child the node is left through
					e.y = lshlOr( ci, 31u, mvrt_f2u( tx1 ) );
					e.z = bfi( 0x7FFFFFFFu, mvrt_f2u( ty1 ), ci << 30 );
					e.w = bfi( 0x7FFFFFFFu, mvrt_f2u( tz1 ), ci << 29 );
					myRing[slot * 64] = e;
					if( !EMBED ) myRingMask[slot * 64] = ( TREE && ( ( s.levelsM1 - level ) & 1u ) ) ? bLo : nodeMask;
					if( TREE ) myRingMask2[slot * 64] = bHi;
					pending |= 1u << level;
					inLds |= 1u << level;
				}
				if( EMBED )
				{
					// :381 -- 32-bit byte offset from the uniform node base (global_load with an SGPR base, no 64-bit VALU adds)
					node = *(const uint32_t*)( (const char*)kids + ( ( ( node & 0xFFFFFFu ) << 5 ) | ( childIndex << 2 ) ) );
				}
				else if( TREE )
				{
					treeDescend( nodes, s.levelsM1, level, childIndex, &node, &nodeMask, &bLo, &bHi, &path );
				}
				else
				{
					const Node64* nd = nodes + node; // up to 2^32 nodes: 64-bit addressing
					nodeMask = ( nd->psum[childIndex >> 2] >> ( 8u * ( childIndex & 3u ) ) ) & 0xFFu; // the child's mask: same line as
					node = nd->children[childIndex];												 // its pointer
				}
				descents++;
				if( !TREE ) path = ( path << 3 ) | childIndex;
				tx1 = mvrt_u2f( bfi( bitMask( ci, 0 ), mvrt_f2u( tx1 ), mvrt_f2u( txM ) ) ); // :382-386: upper half -> keep the exit time, else the mid-plane
				ty1 = mvrt_u2f( bfi( bitMask( ci, 1 ), mvrt_f2u( ty1 ), mvrt_f2u( tyM ) ) );
				tz1 = mvrt_u2f( bfi( bitMask( ci, 2 ), mvrt_f2u( tz1 ), mvrt_f2u( tzM ) ) );
				level++;
			}
			// hit (:324-334) or miss: the lane holds its result until the next refill.  Selects, not a branch: some lane finishes in
			// almost every iteration of a 64-lane wave anyway
			// (a hit only changes the lane's state: t and nMajor are re-derived by finishedHit() from the slab state the lane keeps)
			// bit-sliced child mask of the lanes that popped = the sign bits of the restored exit times; a descent starts a first visit
			// (`popped` is zero for the lanes that did not pop)
			cmX = ( cmX & ~mPopOk ) | signMask( popped.y );
			cmY = ( cmY & ~mPopOk ) | signMask( popped.z );
			cmZ = ( cmZ & ~mPopOk ) | signMask( popped.w );
			mFirst = ( mFirst & ~mPopOk ) | mGo;
			// which lanes are still traversing / hold a hit / hold a miss is kept in lane masks (scalar unit) while the loop runs
			const lmask mMiss = mPop & ~mPopOk;
			hitM |= mHit;
			missM |= mMiss;
			actM &= ~( mHit | mMiss );
			const int nAct = __builtin_popcount( (uint32_t)actM ) + __builtin_popcount( (uint32_t)( actM >> 32 ) ); // (two 32-bit counts: a 64-bit one is compared on the VALU)
			if( nAct == 0 || ( nAct <= 64 - MVRT_REFILL_MIN_OF( FL ) && !exhausted ) ) break;
		}
		st = LANE( hitM ) ? 3u : ( LANE( missM ) ? 2u : st );
		childMask = LANE( mFirst ) ? 8u : ( ( LANE( cmX ) ? 1u : 0u ) | ( LANE( cmY ) ? 2u : 0u ) | ( LANE( cmZ ) ? 4u : 0u ) );
	}
}
