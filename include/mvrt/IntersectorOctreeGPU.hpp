// IntersectorOctreeGPU.hpp -- header-only C++ mirror of the reference's host struct
// (reference IntersectorOctreeGPU.hpp:21-275) over the C-ABI in include/mvrt.h.
//
// Same member names, same call order, same void signatures.  What a reference call site has to change:
//   * glm::vec3 vectors arrive as `const std::vector<mvrt::vec3>&` (any 3-float POD of the same layout,
//     glm::vec3 included -- reinterpret_cast is enough: the reference itself static_asserts
//     sizeof(glm::vec3) == sizeof(float3), IntersectorOctreeGPU.hpp:61);
//   * the `Shader* voxKernel` argument is accepted and ignored (kernels are precompiled into libmvrt_hip.so);
//   * `oroStream` becomes `void*` (a hipStream_t).
// Failures abort(), like the reference (hipUtil.hpp:18-22, IntersectorOctreeGPU.hpp:48-51).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../mvrt.h"

namespace mvrt
{
struct vec3
{
	float x, y, z;
};
inline void check( int rc, const char* what )
{
	if( rc != 0 )
	{
		std::fprintf( stderr, "%s failed: %s\n", what, mvrt_last_error() );
		std::abort();
	}
}

struct IntersectorOctreeGPU
{
	IntersectorOctreeGPU() { check( mvrt_svo_create( &m_handle ), "mvrt_svo_create" ); }
	explicit IntersectorOctreeGPU( mvrt_svo* borrowed ) : m_handle( borrowed ), m_owned( false ) {}
	~IntersectorOctreeGPU()
	{
		if( m_owned && m_handle ) mvrt_svo_destroy( m_handle );
	}
	IntersectorOctreeGPU( const IntersectorOctreeGPU& ) = delete;
	void operator=( const IntersectorOctreeGPU& ) = delete;

	// reference :26-38
	void cleanUp()
	{
		if( m_owned )
		{
			mvrt_svo_destroy( m_handle );
			check( mvrt_svo_create( &m_handle ), "mvrt_svo_create" );
		}
		refresh();
	}

	// reference :40-47 -- build(vertices, vcolors, vemissions, Shader*, stream, origin, dps, gridRes)
	// (buildFlags: MVRT_BUILD_NO_DAG | MVRT_BUILD_NO_EMBEDDED_MASK | MVRT_BUILD_CONSERVATIVE -- the reference's compile-time switches
	// ENABLE_GPU_DAG / ENABLE_EMBEDED_MASK, voxCommon.hpp:5-9, and VTContext's sixSeparating flag, as run-time options; 0 = the reference's defaults)
	template <class V3>
	void build( const std::vector<V3>& vertices, const std::vector<V3>& vcolors, const std::vector<V3>& vemissions, void* /*voxKernel*/, void* stream, V3 origin, float dps,
				int gridRes, int buildFlags = 0 )
	{
		static_assert( sizeof( V3 ) == 3 * sizeof( float ), "vertex type must be 3 packed floats" );
		const float o[3] = { origin.x, origin.y, origin.z };
		check( mvrt_svo_build_ex( m_handle, reinterpret_cast<const float*>( vertices.data() ), vcolors.empty() ? nullptr : reinterpret_cast<const float*>( vcolors.data() ),
								  vemissions.empty() ? nullptr : reinterpret_cast<const float*>( vemissions.data() ), vertices.size(), stream, o, dps, gridRes, buildFlags ),
			   "IntersectorOctreeGPU::build" );
		refresh();
	}

	// adopt an octree built on the CPU (IntersectorOctree::buildDAGReference), reference 68-byte nodes
	void upload( const void* nodes68, uint32_t numberOfNodes, const void* attribs8, uint32_t numberOfVoxels, vec3 origin, float dps, int gridRes, bool hasEmission,
				 bool embeddedMask, void* stream )
	{
		const float o[3] = { origin.x, origin.y, origin.z };
		check( mvrt_svo_upload( m_handle, nodes68, numberOfNodes, attribs8, numberOfVoxels, o, dps, gridRes, hasEmission, embeddedMask, stream ), "mvrt_svo_upload" );
		refresh();
	}

	// batch form of the device method intersect() (:243-251): SoA device arrays
	void intersect( uint64_t n, const float* rox, const float* roy, const float* roz, const float* rdx, const float* rdy, const float* rdz, const uint8_t* isShadowRay, float* t,
					int32_t* nMajor, uint32_t* vIndex, void* stream ) const
	{
		check( mvrt_trace_batch( m_handle, n, rox, roy, roz, rdx, rdy, rdz, isShadowRay, t, nMajor, vIndex, nullptr, stream ), "IntersectorOctreeGPU::intersect" );
	}
	bool hasEmission() const { return m_hasEmission != 0; } // :261-264

	mvrt_svo* handle() const { return m_handle; }

	// re-bind to a handle owned by someone else (PathTracer::m_intersectorOctreeGPU is a VALUE member in the reference, PathTracer.hpp:18)
	void attach( mvrt_svo* borrowed )
	{
		if( m_owned && m_handle ) mvrt_svo_destroy( m_handle );
		m_handle = borrowed;
		m_owned = false;
		refresh();
	}

	// reference public members (:265-274), refreshed after build/upload
	const void* m_vAttributeBuffer = nullptr; // device: VoxelAttirb[m_numberOfVoxels]
	const void* m_nodeBuffer = nullptr;		  // device: 64-byte node lines (see mvrt.h; mvrt_svo_download gives the reference's 68-byte nodes)
	uint32_t m_numberOfNodes = 0;
	uint32_t m_numberOfVoxels = 0;
	vec3 m_lower = { 0, 0, 0 };
	vec3 m_upper = { 0, 0, 0 };
	float m_dps = 0.0f;
	float m_emissionScale = 7.5f;
	uint32_t m_hasEmission = 0;

	void refresh()
	{
		mvrt_svo_info i;
		if( !m_handle || mvrt_svo_get_info( m_handle, &i ) != 0 ) return;
		m_vAttributeBuffer = mvrt_svo_attribute_buffer_dev( m_handle );
		m_nodeBuffer = mvrt_svo_node_buffer_dev( m_handle );
		m_numberOfNodes = i.numberOfNodes;
		m_numberOfVoxels = i.numberOfVoxels;
		m_lower = { i.lower[0], i.lower[1], i.lower[2] };
		m_upper = { i.upper[0], i.upper[1], i.upper[2] };
		m_dps = i.dps;
		m_emissionScale = i.emissionScale;
		m_hasEmission = i.hasEmission;
	}

private:
	mvrt_svo* m_handle = nullptr;
	bool m_owned = true;
};
} // namespace mvrt
