// traverse.h -- per-lane parametric octree traversal for gfx950.
//
// Results are defined by the reference's octreeTraverse_EfficientParametric (voxCommon.hpp:231-423,
// SMALL_STACK variant): same ray mirroring, same slab arithmetic, same child order, same tie-breaks,
// same hit test, so (t, nMajor, vIndex) are bit-identical.  The control structure is NOT the
// reference's: its two nested loops (one per visited node, one per candidate child) are flattened
// into a single loop whose body examines exactly one candidate child and then either descends,
// advances to the node's next candidate, or pops.  On a 64-lane wavefront that removes the
// serialisation of "lanes at the top of the outer loop" against "lanes inside the inner loop":
// every lane executes the same short body with predicated push / pop / fetch.  The invariant that
// makes this legal: inside the reference's inner loop x1 == (childMask & 1 ? cur.tx1 : txM) always
// holds (likewise y, z), so the candidate planes can be recomputed from (cur, childMask) alone.
//
// Stack: LDS, transposed [slot][field][lane] so a wave's push or pop is one conflict-free
// ds_write_b32 / ds_read_b32 per field.  Six dwords per slot: nodeIndex, tx1, ty1, tz1,
// nVoxelSkipped, packed{childMask:3 | level:5 | nodeMask:8}; the reference's `scale` field is
// 2^-level and is rebuilt from the level.  Depth needed = log2(gridRes) (one push per level at most).
#pragma once
#include "mvrt_common.h"

#define MVRT_STACK_FIELDS 6

struct TraceResult
{
	float t;
	int nMajor;
	uint32_t vIndex;
	uint32_t descents;
};

// lds: base of this block's stack area; lane-private column = lds[(slot*6+field)*stride + lane]
template <bool COUNT_DESCENTS>
MVRT_DI TraceResult traceRay( const SvoDev& s, f3 ro, f3 rd, bool isShadowRay, uint32_t* lds, uint32_t stride, uint32_t lane )
{
	TraceResult res;
	res.t = MVRT_MAXF;
	res.nMajor = -1;
	res.vIndex = 0;
	res.descents = 0;

	// voxCommon.hpp:240-260
	float ix = 1.0f / rd.x, iy = 1.0f / rd.y, iz = 1.0f / rd.z;
	uint32_t vMask = 0;
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
	// :265-269
	ix = smin( ix, MVRT_MAXF / smax( smax( sabs( s.lower.x - ro.x ), sabs( s.upper.x - ro.x ) ), 1.0f ) );
	iy = smin( iy, MVRT_MAXF / smax( smax( sabs( s.lower.y - ro.y ), sabs( s.upper.y - ro.y ) ), 1.0f ) );
	iz = smin( iz, MVRT_MAXF / smax( smax( sabs( s.lower.z - ro.z ), sabs( s.upper.z - ro.z ) ), 1.0f ) );
	// :271-278
	float t0x = ( s.lower.x - ro.x ) * ix, t0y = ( s.lower.y - ro.y ) * iy, t0z = ( s.lower.z - ro.z ) * iz;
	float tx1 = ( s.upper.x - ro.x ) * ix, ty1 = ( s.upper.y - ro.y ) * iy, tz1 = ( s.upper.z - ro.z ) * iz;
	if( min3f( tx1, ty1, tz1 ) < max3f( t0x, t0y, t0z ) )
	{
		return res;
	}
	const float dtx = tx1 - t0x, dty = ty1 - t0y, dtz = tz1 - t0z; // :312

	uint32_t node = s.rootIndex;  // pure index (no mask bits)
	uint32_t nodeMask = s.rootMask; // :306
	uint32_t level = 0;			  // scale = 2^-level
	uint32_t childMask = 8u;	  // bit 3 = "not initialised yet" (reference: 0xFFFFFFFF)
	uint32_t skipped = 0;
	uint32_t sp = 0;
	uint32_t* col = lds + lane;

	for( ;; )
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
				res.t = S;
				res.nMajor = ( S == tx0 ) ? 1 : ( ( S == ty0 ) ? 2 : 0 );
				res.vIndex = skipped;
				break;
			}
			pop = true;
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
			const float x1 = ( childMask & 1u ) ? tx1 : txM; // :358-360 and the inner-loop invariant
			const float y1 = ( childMask & 2u ) ? ty1 : tyM;
			const float z1 = ( childMask & 4u ) ? tz1 : tzM;
			const float u = min3f( x1, y1, z1 );								// :365
			const uint32_t mv = ( u == x1 ) ? 1u : ( ( u == y1 ) ? 2u : 4u );	// :366
			const bool hasNext = ( childMask & mv ) == 0;						// :368
			const uint32_t childIndex = childMask ^ vMask;						// :369
			const uint32_t nextMask = childMask | mv;							// :370
			const bool go = ( ( nodeMask >> childIndex ) & 1u ) && !( u < 0.0f ); // :373-375
			if( go )
			{
				if( hasNext ) // :377-380
				{
					uint32_t* p = col + sp * ( MVRT_STACK_FIELDS * stride );
					p[0] = node;
					p[stride] = __float_as_uint( tx1 );
					p[2 * stride] = __float_as_uint( ty1 );
					p[3 * stride] = __float_as_uint( tz1 );
					p[4 * stride] = skipped;
					p[5 * stride] = nextMask | ( level << 3 ) | ( nodeMask << 8 );
					sp++;
				}
				const Node64* nd = s.nodes + node;
				uint32_t child = nd->children[childIndex]; // :381
				if( !isShadowRay )
				{
					skipped += s.embedded ? nd->psum[childIndex] : s.psumCold[(uint64_t)node * 8 + childIndex]; // :388-391
				}
				if( COUNT_DESCENTS ) res.descents++;
				if( s.embedded )
				{
					if( child == MVRT_LEAF )
					{
						node = MVRT_LEAF;
					}
					else
					{
						node = child & 0xFFFFFFu;
						nodeMask = child >> 24;
					}
				}
				else
				{
					node = child;
					nodeMask = ( nd->psum[childIndex >> 2] >> ( 8u * ( childIndex & 3u ) ) ) & 0xFFu; // child masks sit in the parent's line
				}
				tx1 = x1; // :382-386
				ty1 = y1;
				tz1 = z1;
				level++;
				childMask = 8u;
			}
			else if( hasNext ) // :396-411: advance to the next candidate of this node
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
			if( sp == 0 ) break;
			sp--;
			const uint32_t* p = col + sp * ( MVRT_STACK_FIELDS * stride );
			node = p[0];
			tx1 = __uint_as_float( p[stride] );
			ty1 = __uint_as_float( p[2 * stride] );
			tz1 = __uint_as_float( p[3 * stride] );
			skipped = p[4 * stride];
			const uint32_t pk = p[5 * stride];
			childMask = pk & 7u;
			level = ( pk >> 3 ) & 31u;
			nodeMask = ( pk >> 8 ) & 0xFFu;
		}
	}
	return res;
}
