// PathTracer.hpp -- header-only C++ mirror of the reference's PathTracer (reference PathTracer.hpp:14-170)
// over the C-ABI in include/mvrt.h.  Same method names and call order as the reference's callers use
// (voxPTGPU.cpp:88-193, RTCamp.cpp:111-169):
//
//     PathTracer pt;
//     pt.setup( stream, kernelPath, includeDir, isNvidia );      // last three ignored: no runtime compilation
//     pt.resizeFrameBufferIfNeeded( stream, w, h );
//     pt.loadHDRI( stream, "monks_forest_s.hdr", "monks_forest_2k_primary.hdr" );
//     pt.updateScene( vertices, vcolors, vemissions, stream, origin, dps, gridRes );
//     pt.step( stream, viewMatrix, projMatrix, focus, lensR );   // prlib's Camera3D -> the two matrices
//     pt.toImageAsync( stream, rgba8 );                          // pr::Image2DRGBA8 -> uint8_t*
//
// prlib types at the boundary are replaced by plain data: the camera is the (view, proj) pair that
// GetCameraMatrix produces in the reference (PathTracer.hpp:152-153), images are raw pointers.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "IntersectorOctreeGPU.hpp"

namespace mvrt
{
// What the reference's callers touch of hipUtil.hpp:48-74 `Buffer` on the frame buffers (RTCamp.cpp:169 `pt.m_frameBufferU8->data()`): a NON-owning view of
// a device buffer the library owns (it is re-pointed by resizeFrameBufferIfNeeded).
class Buffer
{
public:
	Buffer( void* ptr, int64_t bytes ) : m_bytes( bytes ), m_ptr( ptr ) {}
	int64_t bytes() const { return m_bytes; }
	char* data() { return (char*)m_ptr; }

private:
	int64_t m_bytes;
	void* m_ptr;
};

struct PathTracer
{
	PathTracer() {}
	~PathTracer() { cleanUp(); }
	PathTracer( const PathTracer& ) = delete;
	void operator=( const PathTracer& ) = delete;

	int getSteps() const { return m_handle ? mvrt_pt_get_steps( m_handle ) : 0; }					   // :33
	uint64_t getNumberOfVoxels() const { return m_handle ? mvrt_pt_get_number_of_voxels( m_handle ) : 0; } // :34-37
	uint64_t getOctreeBytes() const { return m_handle ? mvrt_pt_get_octree_bytes( m_handle ) : 0; }	   // :38-41

	// :43-69
	void setup( void* stream, const char* /*kernel*/ = nullptr, const char* /*includeDir*/ = nullptr, bool /*isNvidia*/ = false )
	{
		if( !m_handle )
		{
			check( mvrt_pt_create( &m_handle ), "mvrt_pt_create" );
			m_intersectorOctreeGPU.attach( mvrt_pt_intersector( m_handle ) );
		}
		check( mvrt_pt_setup( m_handle, stream ), "PathTracer::setup" );
	}
	void cleanUp() // :71-79
	{
		m_frameBufferU8.reset();
		m_frameBufferF32.reset();
		m_intersectorOctreeGPU.attach( nullptr );
		if( m_handle ) mvrt_pt_destroy( m_handle );
		m_handle = nullptr;
	}
	void resizeFrameBufferIfNeeded( void* stream, int width, int height ) // :81-97
	{
		check( mvrt_pt_resize_framebuffer_if_needed( m_handle, stream, width, height ), "PathTracer::resizeFrameBufferIfNeeded" );
		m_width = width;
		m_height = height;
		// :90-93 -- uchar4 / float4 per OWNED pixel (= every pixel unless setTile split the frame)
		const int64_t owned = (int64_t)mvrt_pt_owned_pixels( m_handle );
		m_frameBufferU8.reset( new Buffer( mvrt_pt_framebuffer_u8_dev( m_handle ), owned * 4 ) );
		m_frameBufferF32.reset( new Buffer( mvrt_pt_framebuffer_dev( m_handle ), owned * 16 ) );
		m_steps = mvrt_pt_get_steps( m_handle );
	}
	void clearFrameBuffer( void* stream ) // :98-102
	{
		check( mvrt_pt_clear_framebuffer( m_handle, stream ), "PathTracer::clearFrameBuffer" );
		m_steps = 0;
	}
	void loadHDRI( void* stream, const char* file, const char* filePrimary = 0 )													 // :104-116
	{
		check( mvrt_pt_load_hdri_file( m_handle, stream, file, filePrimary ), "PathTracer::loadHDRI" );
	}
	void toImageAsync( void* stream, uint8_t* rgba8 ) { check( mvrt_pt_to_image_async( m_handle, stream, rgba8 ), "PathTracer::toImageAsync" ); } // :118-129
	void resolve( void* stream ) { check( mvrt_pt_resolve( m_handle, stream ), "PathTracer::resolve" ); }										   // :130-137

	template <class V3>
	void updateScene( const std::vector<V3>& vertices, const std::vector<V3>& vcolors, const std::vector<V3>& vemissions, void* stream, V3 origin, float dps, int gridRes ) // :139-148
	{
		m_intersectorOctreeGPU.build( vertices, vcolors, vemissions, nullptr, stream, origin, dps, gridRes );
	}
	// :150-169; view/proj are the column-major 4x4 matrices GetCameraMatrix returns
	void step( void* stream, const float view[16], const float proj[16], float focus, float lensR )
	{
		check( mvrt_pt_step_matrices( m_handle, stream, view, proj, focus, lensR ), "PathTracer::step" );
		m_steps = mvrt_pt_get_steps( m_handle ); // :159 m_steps++
	}

	// multi-GPU extension (not in the reference): render only the 256-pixel blocks b with b % tileCount == tileIndex
	void setTile( int tileIndex, int tileCount ) { check( mvrt_pt_set_tile( m_handle, tileIndex, tileCount ), "PathTracer::setTile" ); }

	mvrt_pt* handle() const { return m_handle; }

	IntersectorOctreeGPU m_intersectorOctreeGPU; // reference member m_intersectorOctreeGPU (:18), a value as there; bound to the handle's octree by setup()
	std::unique_ptr<Buffer> m_frameBufferU8;  // :23 (filled by resolve / toImageAsync); views, see Buffer above
	std::unique_ptr<Buffer> m_frameBufferF32; // :24
	int m_width = 0;
	int m_height = 0;
	int m_steps = 0; // :27

private:
	mvrt_pt* m_handle = nullptr;
};
} // namespace mvrt
