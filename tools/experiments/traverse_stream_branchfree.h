// traverse_stream.h -- persistent-wave octree traversal for gfx950.
//
// Two node-reference flavours (template parameter EMBED):
//   EMBED = true  : child pointers carry the child's occupancy mask in bits 24-31 (ENABLE_EMBEDED_MASK,
//                   voxCommon.hpp:7-9); < 2^24 nodes; node offsets fit 32 bits.
//   EMBED = false : plain 32-bit child indices (up to 2^32-2 nodes, e.g. the 8192^3 non-DAG stress octree).  The
//                   reference reads a node's mask from the node when it is entered (voxCommon.hpp:353-356), i.e. two
//                   dependent misses per descent on an HBM-resident tree.  Here the 8 child masks sit in the parent's
//                   64-byte line next to the 8 child pointers (nVoxelsPSum moves to a cold array that only
//                   voxelIndexFromPath reads), so a descent is ONE line fetch, exactly like the embedded flavour; the
//                   node mask travels in a fifth stack dword.
//
// Same results as traverse.h / the reference's octreeTraverse_EfficientParametric
// (voxCommon.hpp:231-423): identical slab arithmetic, child order, tie-breaks and hit test.  What is
// different is everything the hardware cares about.  Measured on MI355X the traversal is LATENCY
// bound -- throughput is proportional to resident waves per CU (profiles/r01_occupancy_sweep.txt) --
// so this kernel is built around residency and lane utilisation:
//
//  * 16-byte stack entries.  The reference saves 32 bytes per level (voxCommon.hpp:202-212).  Here:
//      - slot index = tree level of the saved node.  Pending entries are ancestors of the current node,
//        hence at strictly increasing levels, so "which entries are pending" is a 32-bit mask in a
//        register; pop = highest set bit.  Neither sp nor the level is stored.
//      - scale = 2^-level is rebuilt from the level.
//      - childMask (3 bits) rides in the sign bits of tx1/ty1/tz1: a saved node was entered with
//        min(x1,y1,z1) >= 0, so its exit times are never negative (a -0.0 would come back as +0.0,
//        which no comparison or output can distinguish).
//      - nVoxelSkipped is not saved at all: the path of child indices (3 bits per level, one 64-bit
//        register = the hit voxel's morton code) is kept instead; the traversal reports that path and
//        the CONSUMER of the hit sums nVoxelsPSum along it (voxelIndexFromPath) in a dense kernel where
//        all 64 lanes walk together.  This also removes the nVoxelsPSum load from every descent.
//    An entry is {child reference (index | mask << 24), tx1, ty1, tz1} = one ds_write_b128.
//  * 4-slot LDS ring per lane (slot = level & 3) = 4 KiB per wave, so 32 waves fit a CU's 160 KiB.
//    A push that lands on an occupied slot first evicts that (shallower) entry to an HBM spill array
//    laid out [level][lane] (coalesced 1 KiB rows); a pop of an evicted level reads it back.  Hot
//    pushes and pops near the leaves never leave LDS.
//  * Persistent waves with lane refill: a wave owns a cursor into the ray stream (grabbed in chunks with
//    one atomic per chunk); whenever at least REFILL_MIN lanes have finished it loads new rays into
//    exactly those lanes.  Long rays no longer hold 63 idle lanes hostage.  Per-ray results do not
//    depend on which lane or wave traced them.
#pragma once
#include "mvrt_common.h"

#define MVRT_RING 4			// LDS ring slots per lane
#ifndef MVRT_REFILL_MIN
#define MVRT_REFILL_MIN 20 // refill once this many lanes are idle (or all of them)
#endif

struct StreamHit
{
	float t;
	int nMajor;
	uint64_t path; // child indices root -> hit voxel, 3 bits per level (= the voxel's morton code); 0 on a miss
	uint32_t descents;
};

// vIndex of the voxel at `path` = sum of nVoxelsPSum along root -> voxel (voxCommon.hpp:388-391).  Done by the
// CONSUMER of a hit (dense kernels, every lane busy), not inside the divergent traversal loop.
MVRT_DI uint32_t voxelIndexFromPath( const SvoDev& s, uint64_t path )
{
	uint32_t n = s.rootIndex, v = 0;
	for( uint32_t l = 0; l < s.levels; l++ )
	{
		const uint32_t c = (uint32_t)( path >> ( 3u * ( s.levels - 1u - l ) ) ) & 7u;
		const Node64* nd = s.nodes + n;
		if( s.embedded )
		{
			v += nd->psum[c];
			n = nd->children[c] & 0xFFFFFFu;
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
	float lox, loy, loz, hix, hiy, hiz;
	uint32_t rootRef; // rootIndex | rootMask << 24 (voxCommon.hpp:306)
	uint32_t rootIndex, rootMask;
};
MVRT_HDI TraceCore makeTraceCore( const SvoDev& s )
{
	TraceCore c;
	c.nodes = s.nodes;
	c.lox = s.lower.x; c.loy = s.lower.y; c.loz = s.lower.z;
	c.hix = s.upper.x; c.hiy = s.upper.y; c.hiz = s.upper.z;
	c.rootRef = s.rootIndex | ( s.rootMask << 24 );
	c.rootIndex = s.rootIndex;
	c.rootMask = s.rootMask;
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
typedef unsigned long long lmask;						   // one bit per lane, wave-uniform (SGPR pair)
#define LANE( m ) __builtin_amdgcn_inverse_ballot_w64( m ) // this lane's bit of a lane mask, as a select condition
// lane select on a lane mask held in an SGPR pair: one v_cndmask_b32, issued as asm so that the optimiser cannot regroup a
// run of selects into an exec-masked block behind a skip branch
MVRT_DI uint32_t selU( lmask m, uint32_t a, uint32_t b ) // lane bit set ? a : b
{
	uint32_t r;
	asm( "v_cndmask_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( b ), "v"( a ), "s"( m ) );
	return r;
}
MVRT_DI uint32_t selUAfter( lmask m, uint32_t a, uint32_t b, float after ) // selU that cannot be scheduled before `after` exists
{
	uint32_t r;
	asm( "v_cndmask_b32 %0, %1, %2, %3" : "=v"( r ) : "v"( b ), "v"( a ), "s"( m ), "v"( after ) );
	return r;
}
MVRT_DI float selF( lmask m, float a, float b ) { return mvrt_u2f( selU( m, mvrt_f2u( a ), mvrt_f2u( b ) ) ); }
MVRT_DI float selAbsF( lmask m, uint32_t aBits, float b ) // lane bit set ? |a| : b   (sign bit of a carries a flag)
{
	float r;
	asm( "v_cndmask_b32 %0, %1, |%2|, %3" : "=v"( r ) : "v"( b ), "v"( aBits ), "s"( m ) );
	return r;
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
template <bool EMBED>
MVRT_DI void traceIrregular( const TraceCore& s, float tx1, float ty1, float tz1, float t0x, float t0y, float t0z, uint32_t vMask, uint4* __restrict__ mySpill,
							 uint64_t spillStride, float* resT, int* resN, uint64_t* pathOut, uint32_t* descentsOut )
{
	const float dtx = tx1 - t0x, dty = ty1 - t0y, dtz = tz1 - t0z;
	uint32_t node = EMBED ? s.rootRef : s.rootIndex, nodeMask = s.rootMask, level = 0, childMask = 8u, sp = 0, descents = 0;
	uint64_t path = 0;
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
					b.y = (uint32_t)path;
					b.z = (uint32_t)( path >> 32 );
					b.w = nodeMask;
					mySpill[(uint64_t)( 2 * sp ) * spillStride] = a;
					mySpill[(uint64_t)( 2 * sp + 1 ) * spillStride] = b;
					sp++;
				}
				if( EMBED )
				{
					node = s.nodes[node & 0xFFFFFFu].children[childIndex];
				}
				else
				{
					const Node64* nd = s.nodes + node;
					nodeMask = ( nd->psum[childIndex >> 2] >> ( 8u * ( childIndex & 3u ) ) ) & 0xFFu; // the child's mask, same line
					node = nd->children[childIndex];
				}
				descents++;
				path = ( path << 3 ) | childIndex;
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
			path = (uint64_t)b.y | ( (uint64_t)b.z << 32 );
			nodeMask = b.w;
		}
	}
	*pathOut = path;
	*descentsOut = descents;
}

// IO concept (ray indices are 32-bit: a launch never exceeds 2^32 rays):
//   bool load( uint32_t ray, f3* ro, f3* rd )   -> returns isShadowRay
//   void store( uint32_t ray, const StreamHit& h, bool isShadowRay )
//
// One loop iteration = (1) refill when enough lanes are idle: first STORE the results those lanes still hold
// (one store site, many lanes per store instruction), then load new rays into them; (2) one traversal step for
// every active lane: straight-line bit arithmetic (v_bfi / v_bfe selects instead of compare-select chains)
// followed by three shallow branches: descend (with push), pop, hit.
template <bool EMBED, class IO>
MVRT_DI void traceStream( const TraceCore& s, IO& io, uint64_t total64, unsigned long long* __restrict__ cursor, uint32_t chunk, uint4* __restrict__ ldsRing /* [MVRT_RING][64] */,
						  uint4* __restrict__ spill /* [levels][spillStride] */, uint64_t spillStride, uint64_t spillLane, uint32_t* __restrict__ ldsMask = nullptr /* [MVRT_RING][64], !EMBED */,
						  uint32_t* __restrict__ spillMask = nullptr /* [levels][spillStride], !EMBED */ )
{
	const uint32_t lane = threadIdx.x;
	const uint32_t total = (uint32_t)total64;
	const Node64* __restrict__ nodes = s.nodes;
	// LDS ring: explicit LDS address space (so the optimiser cannot fold a ring read and a spill read into one flat
	// load); slot k of this lane at byte offset k * 1024 + lane * 16
	LdsU4* const myRing = (LdsU4*)ldsRing + lane;
	const uint32_t ringAddr = (uint32_t)(uintptr_t)myRing; // LDS byte address of this lane's slot 0
	const uint32_t ringMaskAddr = EMBED ? 0u : (uint32_t)(uintptr_t)( (LdsU32*)ldsMask + lane );
	uint4* const mySpill = spill + spillLane; // level L at mySpill[L * spillStride]  (irregular rays only)
	LdsU32* const myRingMask = EMBED ? nullptr : (LdsU32*)ldsMask + lane;
	// spill rows: spillStride is a power of two (traceWorkspaceLanes), rows * stride * 16 B < 4 GiB: row L of this lane
	// is base + ((L << spillShift) + lane offset) with 32-bit arithmetic and a scalar base
	const uint32_t spillShift = 4u + (uint32_t)__builtin_ctzll( spillStride );
	const uint32_t spillOff = (uint32_t)spillLane * 16u;
	const uint32_t spillMaskShift = spillShift - 2u, spillMaskOff = (uint32_t)spillLane * 4u;

	// wave-uniform cursor state
	uint32_t chunkNext = 0, chunkEnd = 0;
	bool exhausted = false;

	// per-lane state.  st: 0 idle, 1 traversing, 2 finished and holding a result that is not stored yet
	uint32_t st = 0;
	bool isShadow = false;
	uint32_t ray = 0;
	float dtx = 0, dty = 0, dtz = 0, tx1 = 0, ty1 = 0, tz1 = 0;
	uint32_t vMask = 0, vMaskHi = 24u, node = 0, nodeMask = 0, level = 0, childMask = 8u, pending = 0, inLds = 0, descents = 0;
	uint64_t path = 0;
	float resT = MVRT_MAXF;
	int resN = -1;

	for( ;; )
	{
		// ---------------- (1) refill: control only gets here when enough lanes are idle ----------------
		const unsigned long long idleMask = __ballot( st != 1u );
		const uint32_t nIdle = __popcll( idleMask );
		{
			if( st == 2u ) // flush results of the lanes that finished since the last refill
			{
				StreamHit h;
				h.t = resT;
				h.nMajor = resN;
				h.path = resT != MVRT_MAXF ? path : 0ull;
				h.descents = descents;
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
						base = __shfl( base, 0, 64 );
						if( base >= total )
						{
							exhausted = true;
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
						isShadow = io.load( ray, &ro, &rd );
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
						resT = MVRT_MAXF;
						resN = -1;
						vMaskHi = vMask | 24u;
						if( min3f( tx1, ty1, tz1 ) < max3f( t0x, t0y, t0z ) ) // :275-278 misses the root box
						{
							st = 2u;
						}
						else if( ( ( mvrt_f2u( tx1 - t0x ) & 0x7F800000u ) == 0x7F800000u ) || ( ( mvrt_f2u( ty1 - t0y ) & 0x7F800000u ) == 0x7F800000u ) ||
								 ( ( mvrt_f2u( tz1 - t0z ) & 0x7F800000u ) == 0x7F800000u ) )
						{
							// irregular ray (inf / NaN slab delta): exact reference emulation, see traceIrregular
							traceIrregular<EMBED>( s, tx1, ty1, tz1, t0x, t0y, t0z, vMask, mySpill, spillStride, &resT, &resN, &path, &descents );
							st = 2u;
						}
						else
						{
							dtx = tx1 - t0x;
							dty = ty1 - t0y;
							dtz = tz1 - t0z;
							node = EMBED ? s.rootRef : s.rootIndex;
							nodeMask = s.rootMask;
							level = 0;
							childMask = 8u;
							pending = 0;
							inLds = 0;
							st = 1u;
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
				if( st == 2u )			   // final flush
				{
					StreamHit h;
					h.t = resT;
					h.nMajor = resN;
					h.path = resT != MVRT_MAXF ? path : 0ull;
					h.descents = descents;
					io.store( ray, h, isShadow );
				}
				break; // every lane idle and the stream is empty: the wave retires
			}
		}

#ifdef MVRT_UTIL_STATS
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
		// BRANCH-FREE step.  On gfx950 a branch instruction -- taken or not, s_cbranch_execz included -- costs the SIMD
		// about as much issue time as four VALU instructions (tools/calib/issue_rate.hip), and the compiler wraps every
		// divergent `if` that contains a memory instruction in one.  So the step has no `if`: every lane executes the
		// same instruction stream; the three outcomes (descend [+push] / pop / hit) are lane predicates that drive
		// selects, and the memory instructions run under an explicit exec mask (inline asm) or on a harmless address.
		//  * push is WRITE-THROUGH: the entry goes to its LDS ring slot and to its spill row in one masked pair of
		//    instructions, so a ring slot can be overwritten without first evicting what it held (no clash test);
		//  * pop reads the ring slot of the deepest pending level for every lane, then re-loads the lanes whose entry
		//    has been overwritten from the spill row under an exec mask.  That load carries sc0 sc1 (served by L2): a
		//    64-byte line of a spill row holds the entries of four neighbouring lanes, and a copy of it in the CU's L1
		//    -- fetched for one lane -- would not see what the other three have stored since;
		//  * the child-pointer load of this step is consumed at the top of the NEXT step (after the slab arithmetic,
		//    which does not depend on it), so its latency overlaps with that arithmetic.
		uint32_t loadedPrev = 0, loadedPrev2 = 0, maskShiftPrev = 0;
		lmask mGoPrev = 0, mActive = __ballot( st == 1u );
		for( ;; )
		{
			// y and z ride in one register pair so that the slab arithmetic issues as packed fp32 (v_pk_mul_f32 / v_pk_add_f32:
			// two IEEE operations per issue slot, each rounded exactly like its scalar twin; no contraction)
			const float scale = mvrt_u2f( ( 127u - level ) << 23 );
			const v2f t1yz = { ty1, tz1 };
			const v2f dtyz = { dty, dtz };
			const float tx0 = tx1 - dtx * scale; // :317-320
			const v2f t0yz = t1yz - dtyz * scale;
			const float ty0 = t0yz.x, tz0 = t0yz.y;
			// no NaNs can reach the decisions of an active lane (the direction clamp keeps every product finite), so max3/min3
			// equal the reference's compare-select chains up to the sign of a zero, which no decision below can see
			const float S = fmaxf( fmaxf( tx0, ty0 ), tz0 );
			const float txM = 0.5f * ( tx0 + tx1 ); // :338-340
			const v2f tMyz = ( t0yz + t1yz ) * 0.5f;
			const float tyM = tMyz.x, tzM = tMyz.y;
			// first visit: childMask bit = (tM < S) = sign bit of (tM - S)  (:342-348; the difference of two
			// distinct floats is never zero with denormals on, and x - x = +0)
			const v2f dMyz = tMyz - S;
			const uint32_t cmInit = ( mvrt_f2u( txM - S ) >> 31 ) | ( ( mvrt_f2u( dMyz.x ) >> 30 ) & 2u ) | ( ( mvrt_f2u( dMyz.y ) >> 29 ) & 4u );
			const uint32_t cm = bfi( bitMask( childMask, 3 ), cmInit, childMask ); // childMask is either 8 (first visit) or a 3-bit mask
			const float x1 = mvrt_u2f( bfi( bitMask( cm, 0 ), mvrt_f2u( tx1 ), mvrt_f2u( txM ) ) ); // :358-360
			const float y1 = mvrt_u2f( bfi( bitMask( cm, 1 ), mvrt_f2u( ty1 ), mvrt_f2u( tyM ) ) );
			const float z1 = mvrt_u2f( bfi( bitMask( cm, 2 ), mvrt_f2u( tz1 ), mvrt_f2u( tzM ) ) );
			const float u = fminf( fminf( x1, y1 ), z1 );					  // :365
			const uint32_t mv = ( u == x1 ) ? 1u : ( ( u == y1 ) ? 2u : 4u ); // :366
			const uint32_t childBit = cm ^ vMaskHi;							  // :369, child index + 24 (vMaskHi = vMask | 24)
			const uint32_t childIndex = childBit & 7u;
			const uint32_t nextMask = cm | mv;								  // :370
			// the child pointer fetched by the previous step lands here (:381)
			// (the select is tied to `u` so that the scheduler cannot hoist it -- and the wait for the load -- above the arithmetic)
			node = selUAfter( mGoPrev, loadedPrev, node, u );
			if( !EMBED ) nodeMask = selUAfter( mGoPrev, ( loadedPrev2 >> maskShiftPrev ) & 0xFFu, nodeMask, u );
			// lane predicates as 64-bit lane masks: the logic runs on the scalar unit and every select below takes its mask
			// straight from an SGPR pair (written with && / ?: on bools it comes back as short-circuit branches)
			const lmask mLeaf = __ballot( node == MVRT_LEAF );	 // :322
			const lmask mHasNext = __ballot( ( cm & mv ) == 0u ); // :368
			const lmask mExists = __ballot( EMBED ? bitMask( node, childBit ) != 0u : ( ( nodeMask >> childIndex ) & 1u ) != 0u );
			const lmask mGo = mActive & ~mLeaf & mExists & ~__ballot( u < 0.0f ); // :373-375
			const lmask mHit = mActive & mLeaf & __ballot( 0.0f < S );			   // :324
			const lmask mPop = mActive & ~mHit & ~mGo & ( mLeaf | ~mHasNext );
			const lmask mAdv = mActive & ~mLeaf & ~mGo & mHasNext;
			childMask = selU( mAdv, nextMask, cm ); // advance within the node (:396-411): only the child mask changes

			// ---- pop (:414-422) ----
			lmask mMissOut;
			{
				const uint32_t L = ( 31u - (uint32_t)__clz( (int)pending ) ) & 31u; // deepest pending level (any value when none)
				const uint32_t bit = 1u << L;
				const lmask mPending = __ballot( pending != 0u );
				const lmask mPopOk = mPop & mPending, mMiss = mPop & ~mPending;
				const lmask mSpill = mPopOk & __ballot( ( inLds & bit ) == 0u );
				const bool popOk = LANE( mPopOk );
				mMissOut = mMiss;
				u4v e;
				uint32_t eMask = 0;
				unsigned long long sv;
				if( EMBED )
				{
					asm volatile( "ds_read_b128 %0, %2\n\t"
								  "s_and_saveexec_b64 %1, %3\n\t"
								  "s_waitcnt lgkmcnt(0)\n\t"
								  "s_cbranch_execz .Lnospill%=\n\t" // rare path: skipped by a branch (an exec = 0 load would still cost a round trip)
								  "global_load_dwordx4 %0, %4, %5 sc0 sc1\n\t"
								  "s_waitcnt vmcnt(0)\n"
								  ".Lnospill%=:\n\t"
								  "s_mov_b64 exec, %1"
								  : "=&v"( e ), "=&s"( sv )
								  : "v"( ringAddr + ( ( L & ( MVRT_RING - 1 ) ) << 10 ) ), "s"( mSpill ), "v"( ( L << spillShift ) + spillOff ), "s"( spill )
								  : "memory", "scc" );
				}
				else
				{
					asm volatile( "ds_read_b128 %0, %3\n\t"
								  "ds_read_b32 %1, %4\n\t"
								  "s_and_saveexec_b64 %2, %5\n\t"
								  "s_waitcnt lgkmcnt(0)\n\t"
								  "s_cbranch_execz .Lnospill%=\n\t"
								  "global_load_dwordx4 %0, %6, %7 sc0 sc1\n\t"
								  "global_load_dword %1, %8, %9 sc0 sc1\n\t"
								  "s_waitcnt vmcnt(0)\n"
								  ".Lnospill%=:\n\t"
								  "s_mov_b64 exec, %2"
								  : "=&v"( e ), "=&v"( eMask ), "=&s"( sv )
								  : "v"( ringAddr + ( ( L & ( MVRT_RING - 1 ) ) << 10 ) ), "v"( ringMaskAddr + ( ( L & ( MVRT_RING - 1 ) ) << 8 ) ), "s"( mSpill ),
									"v"( ( L << spillShift ) + spillOff ), "s"( spill ), "v"( ( L << spillMaskShift ) + spillMaskOff ), "s"( spillMask )
								  : "memory", "scc" );
				}
				const uint32_t cmR = andOr( e.w >> 29, 4u, andOr( e.z >> 30, 2u, e.y >> 31 ) );
				st = selU( mMiss | mHit, 2u, st ); // miss (the stack is empty) or hit: the lane holds its result until the next refill
				node = selU( mPopOk, e.x, node );
				if( !EMBED ) nodeMask = selU( mPopOk, eMask, nodeMask );
				childMask = selU( mPopOk, cmR, childMask );
				tx1 = selAbsF( mPopOk, e.y, tx1 );
				ty1 = selAbsF( mPopOk, e.z, ty1 );
				tz1 = selAbsF( mPopOk, e.w, tz1 );
				const uint32_t up = selU( mPopOk, 3u * ( level - L ), 0u );
				path >>= up;
				level = selU( mPopOk, L, level );
				const uint32_t clr = popOk ? bit : 0u;
				pending &= ~clr;
				inLds &= ~clr;
			}

			// ---- descend (:377-391), with a write-through push when the node has further candidates ----
			{
				const lmask mPush = mGo & mHasNext;
				const bool push = LANE( mPush ), go = LANE( mGo );
				const uint32_t slot = level & ( MVRT_RING - 1 );
				u4v en;
				en.x = node;
				// sign bits of a saved node's exit times are clear (entered with min >= 0; +0 - x and 0.5*(a+b) never
				// yield -0.0 from non-negative-zero inputs), so they can carry the child mask
				en.y = lshlOr( nextMask, 31u, mvrt_f2u( tx1 ) );
				en.z = bfi( 0x7FFFFFFFu, mvrt_f2u( ty1 ), nextMask << 30 );
				en.w = bfi( 0x7FFFFFFFu, mvrt_f2u( tz1 ), nextMask << 29 );
				unsigned long long sv;
				if( EMBED )
				{
					asm volatile( "s_and_saveexec_b64 %0, %1\n\t"
								  "ds_write_b128 %2, %3\n\t"
								  "global_store_dwordx4 %4, %3, %5\n\t"
								  "s_mov_b64 exec, %0\n\t"
								  "s_nop 1" // gfx940+: a VALU write of the data VGPRs of a >8-byte store needs 2 wait states; the
											// compiler cannot see the store inside this block, so the padding is written here
								  : "=&s"( sv )
								  : "s"( mPush ), "v"( ringAddr + ( slot << 10 ) ), "v"( en ), "v"( ( level << spillShift ) + spillOff ), "s"( spill )
								  : "memory", "scc" );
				}
				else
				{
					asm volatile( "s_and_saveexec_b64 %0, %1\n\t"
								  "ds_write_b128 %2, %3\n\t"
								  "ds_write_b32 %4, %5\n\t"
								  "global_store_dwordx4 %6, %3, %7\n\t"
								  "global_store_dword %8, %5, %9\n\t"
								  "s_mov_b64 exec, %0\n\t"
								  "s_nop 1"
								  : "=&s"( sv )
								  : "s"( mPush ), "v"( ringAddr + ( slot << 10 ) ), "v"( en ), "v"( ringMaskAddr + ( slot << 8 ) ), "v"( nodeMask ),
									"v"( ( level << spillShift ) + spillOff ), "s"( spill ), "v"( ( level << spillMaskShift ) + spillMaskOff ), "s"( spillMask )
								  : "memory", "scc" );
				}
				const uint32_t lb = push ? 1u << level : 0u;
				pending |= lb;
				inLds = bfi( push ? ( 0x11111111u << slot ) : 0u, lb, inLds ); // the slot now holds level `level` and nothing shallower
				// child pointer (and, non-embedded, the child masks that sit in the same line); lanes that do not descend read
				// the first dword of the node array
				if( EMBED )
				{
					const uint32_t off = go ? ( ( ( node & 0xFFFFFFu ) << 6 ) | ( childIndex << 2 ) ) : 0u;
					loadedPrev = *(const uint32_t*)( (const char*)nodes + off ); // 32-bit offset from the uniform base
				}
				else
				{
					const uint64_t off = go ? ( (uint64_t)node << 6 ) : 0ull; // up to 2^32 nodes: 64-bit addressing
					const Node64* nd = (const Node64*)( (const char*)nodes + off );
					loadedPrev = nd->children[childIndex];
					loadedPrev2 = nd->psum[childIndex >> 2]; // 4 child masks; the right byte is picked when the pointer is consumed
					maskShiftPrev = 8u * ( childIndex & 3u );
				}
				const uint32_t one = selU( mGo, 1u, 0u );
				descents += one;
				path = ( path << ( 3u * one ) ) | selU( mGo, childIndex, 0u );
				tx1 = selF( mGo, x1, tx1 ); // :382-386
				ty1 = selF( mGo, y1, ty1 );
				tz1 = selF( mGo, z1, tz1 );
				level += one;
				childMask = selU( mGo, 8u, childMask );
			}

			// ---- hit (:324-334) ----
			const int nmY = ( S == ty0 ) ? 2 : 0;
			const int nm = ( S == tx0 ) ? 1 : nmY;
			resT = selF( mHit, S, resT );
			resN = (int)selU( mHit, (uint32_t)nm, (uint32_t)resN );

			mGoPrev = mGo;
			mActive &= ~( mMissOut | mHit );
			const int nDone = 64 - __builtin_popcountll( mActive );
			if( nDone == 64 || ( nDone >= MVRT_REFILL_MIN && !exhausted ) ) break;
		}
		// the last step's child pointer is still in flight for the lanes that descended
		node = LANE( mGoPrev ) ? loadedPrev : node;
		if( !EMBED ) nodeMask = LANE( mGoPrev ) ? ( ( loadedPrev2 >> maskShiftPrev ) & 0xFFu ) : nodeMask;
	}
}
