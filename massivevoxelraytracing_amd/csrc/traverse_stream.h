// traverse_stream.h -- persistent-wave octree traversal for gfx950 (embedded-mask octrees).
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
#define MVRT_REFILL_MIN 20	// refill once this many lanes are idle (or all of them)

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
		v += nd->psum[c];
		n = nd->children[c] & 0xFFFFFFu;
	}
	return v;
}

// IO concept:
//   bool load( uint64_t ray, f3* ro, f3* rd )   -> returns isShadowRay
//   void store( uint64_t ray, const StreamHit& h, bool isShadowRay )
template <class IO>
MVRT_DI void traceStream( const SvoDev& s, IO& io, uint64_t total, unsigned long long* __restrict__ cursor, uint32_t chunk, uint4* __restrict__ ldsRing /* [MVRT_RING][64] */,
						  uint4* __restrict__ spill /* [levels][spillStride] */, uint64_t spillStride, uint64_t spillLane )
{
	const uint32_t lane = threadIdx.x;
	const Node64* __restrict__ nodes = s.nodes;

	// wave-uniform cursor state
	uint64_t chunkNext = 0, chunkEnd = 0;
	bool exhausted = false;

	// per-lane ray state
	bool active = false;
	bool isShadow = false;
	uint64_t ray = 0;
	float dtx = 0, dty = 0, dtz = 0, tx1 = 0, ty1 = 0, tz1 = 0;
	uint32_t vMask = 0, node = 0, level = 0, childMask = 8u, pending = 0, inLds = 0, descents = 0;
	uint64_t path = 0;

	for( ;; )
	{
		// ---------------- refill ----------------
		const unsigned long long idleMask = __ballot( !active );
		const uint32_t nIdle = __popcll( idleMask );
		if( nIdle == 64 || ( nIdle >= MVRT_REFILL_MIN && !exhausted ) )
		{
			if( !exhausted )
			{
				// hand out rays [chunkNext, ...) to idle lanes in lane order; grab new chunks as needed
				uint32_t need = nIdle;
				uint32_t myRank = __popcll( idleMask & ( ( 1ull << lane ) - 1ull ) );
				uint32_t given = 0;
				while( need > 0 )
				{
					if( chunkNext == chunkEnd )
					{
						unsigned long long base = 0;
						if( lane == 0 ) base = atomicAdd( cursor, (unsigned long long)chunk );
						base = __shfl( base, 0, 64 );
						if( base >= total )
						{
							exhausted = true;
							break;
						}
						chunkNext = base;
						chunkEnd = base + chunk < total ? base + chunk : total;
					}
					uint32_t avail = (uint32_t)( chunkEnd - chunkNext );
					uint32_t take = avail < need ? avail : need;
					if( !active && myRank >= given && myRank < given + take )
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
							ro.x = s.lower.x + s.upper.x - ro.x;
						}
						if( iy < 0.0f )
						{
							vMask |= 2u;
							iy = -iy;
							ro.y = s.lower.y + s.upper.y - ro.y;
						}
						if( iz < 0.0f )
						{
							vMask |= 4u;
							iz = -iz;
							ro.z = s.lower.z + s.upper.z - ro.z;
						}
						ix = smin( ix, MVRT_MAXF / smax( smax( sabs( s.lower.x - ro.x ), sabs( s.upper.x - ro.x ) ), 1.0f ) );
						iy = smin( iy, MVRT_MAXF / smax( smax( sabs( s.lower.y - ro.y ), sabs( s.upper.y - ro.y ) ), 1.0f ) );
						iz = smin( iz, MVRT_MAXF / smax( smax( sabs( s.lower.z - ro.z ), sabs( s.upper.z - ro.z ) ), 1.0f ) );
						const float t0x = ( s.lower.x - ro.x ) * ix, t0y = ( s.lower.y - ro.y ) * iy, t0z = ( s.lower.z - ro.z ) * iz;
						tx1 = ( s.upper.x - ro.x ) * ix;
						ty1 = ( s.upper.y - ro.y ) * iy;
						tz1 = ( s.upper.z - ro.z ) * iz;
						descents = 0;
						if( min3f( tx1, ty1, tz1 ) < max3f( t0x, t0y, t0z ) ) // :275-278 miss the root box
						{
							StreamHit h;
							h.t = MVRT_MAXF;
							h.nMajor = -1;
							h.path = 0;
							h.descents = 0;
							io.store( ray, h, isShadow );
						}
						else
						{
							dtx = tx1 - t0x;
							dty = ty1 - t0y;
							dtz = tz1 - t0z;
							node = s.rootIndex | ( s.rootMask << 24 ); // :306
							level = 0;
							childMask = 8u;
							pending = 0;
							inLds = 0;
							path = 0;
							active = true;
						}
					}
					given += take;
					need -= take;
					chunkNext += take;
				}
			}
			if( __ballot( active ) == 0ull )
			{
				if( exhausted ) break; // every lane idle and the stream is empty: the wave retires
				continue;			   // all the rays just loaded missed the root box: fetch again
			}
		}

		// ---------------- one traversal step for every active lane ----------------
		if( active )
		{
			const float scale = mvrt_u2f( ( 127u - level ) << 23 );
			const float tx0 = tx1 - dtx * scale; // :317-320
			const float ty0 = ty1 - dty * scale;
			const float tz0 = tz1 - dtz * scale;
			const float S = max3f( tx0, ty0, tz0 );
			bool pop = false;
			if( node == MVRT_LEAF ) // :322-336
			{
				if( 0.0f < S )
				{
					StreamHit h;
					h.t = S;
					h.nMajor = ( S == tx0 ) ? 1 : ( ( S == ty0 ) ? 2 : 0 );
					h.path = path; // all voxels sit at depth s.levels, so level == s.levels here
					h.descents = descents;
					io.store( ray, h, isShadow );
					active = false;
				}
				else
				{
					pop = true;
				}
			}
			else
			{
				const float txM = 0.5f * ( tx0 + tx1 ); // :338-340
				const float tyM = 0.5f * ( ty0 + ty1 );
				const float tzM = 0.5f * ( tz0 + tz1 );
				if( childMask & 8u ) // :342-348
				{
					childMask = ( txM < S ? 1u : 0u ) | ( tyM < S ? 2u : 0u ) | ( tzM < S ? 4u : 0u );
				}
				const float x1 = ( childMask & 1u ) ? tx1 : txM; // :358-360
				const float y1 = ( childMask & 2u ) ? ty1 : tyM;
				const float z1 = ( childMask & 4u ) ? tz1 : tzM;
				const float u = min3f( x1, y1, z1 );							  // :365
				const uint32_t mv = ( u == x1 ) ? 1u : ( ( u == y1 ) ? 2u : 4u ); // :366
				const bool hasNext = ( childMask & mv ) == 0;					  // :368
				const uint32_t childIndex = childMask ^ vMask;					  // :369
				const uint32_t nextMask = childMask | mv;						  // :370
				const bool go = ( ( node >> ( 24u + childIndex ) ) & 1u ) && !( u < 0.0f );
				if( go )
				{
					if( hasNext ) // push (:377-380)
					{
						const uint32_t slot = level & ( MVRT_RING - 1 );
						const uint32_t clash = inLds & ( 0x11111111u << slot );
						if( clash ) // the slot still holds a shallower pending entry: evict it to HBM
						{
							const uint32_t lc = __builtin_ctz( clash );
							spill[(uint64_t)lc * spillStride + spillLane] = ldsRing[slot * 64 + lane];
							inLds &= ~clash;
						}
						uint4 e;
						e.x = node;
						e.y = ( __float_as_uint( tx1 ) & 0x7FFFFFFFu ) | ( ( nextMask & 1u ) << 31 );
						e.z = ( __float_as_uint( ty1 ) & 0x7FFFFFFFu ) | ( ( nextMask & 2u ) << 30 );
						e.w = ( __float_as_uint( tz1 ) & 0x7FFFFFFFu ) | ( ( nextMask & 4u ) << 29 );
						ldsRing[slot * 64 + lane] = e;
						pending |= 1u << level;
						inLds |= 1u << level;
					}
					node = nodes[node & 0xFFFFFFu].children[childIndex]; // :381
					descents++;
					path = ( path << 3 ) | childIndex;
					tx1 = x1; // :382-386
					ty1 = y1;
					tz1 = z1;
					level++;
					childMask = 8u;
				}
				else if( hasNext ) // :396-411
				{
					childMask = nextMask;
				}
				else
				{
					pop = true;
				}
			}
			if( pop ) // :414-422
			{
				if( pending == 0 )
				{
					StreamHit h;
					h.t = MVRT_MAXF;
					h.nMajor = -1;
					h.path = 0;
					h.descents = descents;
					io.store( ray, h, isShadow );
					active = false;
				}
				else
				{
					const uint32_t L = 31u - __builtin_clz( pending );
					const uint32_t bit = 1u << L;
					uint4 e;
					if( inLds & bit ) e = ldsRing[( L & ( MVRT_RING - 1 ) ) * 64 + lane];
					else e = spill[(uint64_t)L * spillStride + spillLane];
					pending &= ~bit;
					inLds &= ~bit;
					path >>= 3u * ( level - L );
					level = L;
					node = e.x;
					childMask = ( e.y >> 31 ) | ( ( e.z >> 31 ) << 1 ) | ( ( e.w >> 31 ) << 2 );
					tx1 = __uint_as_float( e.y & 0x7FFFFFFFu );
					ty1 = __uint_as_float( e.z & 0x7FFFFFFFu );
					tz1 = __uint_as_float( e.w & 0x7FFFFFFFu );
				}
			}
		}
	}
}
