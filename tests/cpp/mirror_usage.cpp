// Compile-check of the header-only C++ mirrors (include/mvrt/*.hpp): the call sequence of the reference's
// voxPTGPU.cpp:88-193 / voxRTGPU.cpp:104-203 written against them.  Built by tests/test_cpp_mirror.py.
#include <cstdio>
#include <vector>

#include "mvrt/PathTracer.hpp"

struct V3 // stand-in for glm::vec3
{
	float x, y, z;
};

int main( int argc, char** argv )
{
	if( argc < 2 ) // never executed by the CPU test: needs a GPU
	{
		std::printf( "usage: mirror_usage run\n" );
		return 0;
	}
	void* stream = nullptr;
	mvrt::check( mvrt_stream_create( &stream ), "stream" );
	std::vector<V3> vertices = { { 0.1f, 0.1f, 0.1f }, { 0.9f, 0.2f, 0.3f }, { 0.4f, 0.8f, 0.7f } }, vcolors, vemissions;
	mvrt::PathTracer pt;
	pt.setup( stream, "../voxKernel.cu", "../", false );
	pt.resizeFrameBufferIfNeeded( stream, 64, 36 );
	pt.loadHDRI( stream, "monks_forest_s.hdr" );
	pt.updateScene( vertices, vcolors, vemissions, stream, V3{ 0, 0, 0 }, 1.0f / 64, 64 );
	const float view[16] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, -0.5f, -0.5f, -3, 1 };
	const float proj[16] = { 1.3f, 0, 0, 0, 0, 2.4f, 0, 0, 0, 0, -1, -1, 0, 0, -0.2f, 0 };
	pt.clearFrameBuffer( stream );
	pt.step( stream, view, proj, 3.0f, 0.01f );
	std::vector<uint8_t> image( 64 * 36 * 4 );
	pt.toImageAsync( stream, image.data() );
	mvrt::check( mvrt_stream_synchronize( stream ), "sync" );
	std::printf( "steps %d voxels %llu octree bytes %llu\n", pt.getSteps(), (unsigned long long)pt.getNumberOfVoxels(), (unsigned long long)pt.getOctreeBytes() );
	// RTCamp.cpp:160-169: resolve, then a device-to-device copy out of pt.m_frameBufferU8->data() into a pooled buffer
	pt.resolve( stream );
	void* pooled = nullptr;
	mvrt::check( mvrt_malloc( &pooled, 64 * 36 * 4 ), "malloc" );
	mvrt::check( mvrt_memcpy_d2d( pooled, pt.m_frameBufferU8->data(), 64 * 36 * 4, stream ), "d2d" );
	std::vector<uint8_t> copy( 64 * 36 * 4 );
	mvrt::check( mvrt_memcpy_d2h( copy.data(), pooled, copy.size(), stream ), "d2h" );
	std::vector<float> f32( 64 * 36 * 4 );
	mvrt::check( mvrt_memcpy_d2h( f32.data(), pt.m_frameBufferF32->data(), f32.size() * 4, stream ), "d2h" );
	double w = 0;
	for( size_t i = 3; i < f32.size(); i += 4 ) w += f32[i];
	std::printf( "m_steps %d u8 bytes %lld f32 bytes %lld same %d sumW %.1f\n", pt.m_steps, (long long)pt.m_frameBufferU8->bytes(), (long long)pt.m_frameBufferF32->bytes(),
				 (int)( copy == image ), w );
	mvrt_free( pooled );

	mvrt::IntersectorOctreeGPU isect; // voxRTGPU.cpp:104-171
	isect.build( vertices, vcolors, vemissions, nullptr, stream, V3{ 0, 0, 0 }, 1.0f / 64, 64 );
	std::printf( "nodes %u voxels %u hasEmission %d\n", isect.m_numberOfNodes, isect.m_numberOfVoxels, (int)isect.hasEmission() );
	pt.cleanUp();
	return 0;
}
