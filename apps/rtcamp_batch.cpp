// rtcamp_batch.cpp -- batch animation renderer on top of the drop-in boundary, following the structure of the
// reference's RTCamp.cpp: `--frame-range a b` sharding across machines (RTCamp.cpp:47-52 and
// usecase2_submission/run.py), per-frame SVO rebuild with a voxel-resolution ramp rounded up to a power of two
// (:136-142), clearFrameBuffer + N x step + resolve per frame (:153-166), and a worker thread that reads frames back
// through a 4-buffer pool and writes image files while the GPU renders the next frame (:126-130,165-193).
//
// The Alembic scene / camera animation of the reference (prlib, absent) is replaced by a Wavefront .obj mesh and an
// orbiting look-at camera.  Everything GPU-side goes through include/mvrt/PathTracer.hpp -> libmvrt_hip.so.
//
//   rtcamp_batch scene.obj env.hdr outdir [--frame-range a b] [--frames N] [--size W H] [--res from to] [--steps K] [--png]
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mvrt/PathTracer.hpp"
#include "scene_io.hpp"

static uint32_t next_power_of_two( uint32_t v ) // RTCamp.cpp:13-23
{
	v--;
	v |= v >> 1;
	v |= v >> 2;
	v |= v >> 4;
	v |= v >> 8;
	v |= v >> 16;
	v++;
	return v;
}

// look-at view matrix + GL perspective, column-major (what prlib's GetCameraMatrix hands to PathTracer::step)
static void lookAt( const float eye[3], const float at[3], float fovyDeg, float aspect, float view[16], float proj[16] )
{
	float f[3] = { at[0] - eye[0], at[1] - eye[1], at[2] - eye[2] };
	float fl = std::sqrt( f[0] * f[0] + f[1] * f[1] + f[2] * f[2] );
	for( float& c : f ) c /= fl;
	float up[3] = { 0, 1, 0 };
	float s[3] = { f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0] };
	float sl = std::sqrt( s[0] * s[0] + s[1] * s[1] + s[2] * s[2] );
	for( float& c : s ) c /= sl;
	float u[3] = { s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0] };
	const float m[16] = { s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
						  -( s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2] ), -( u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2] ), f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2], 1 };
	std::memcpy( view, m, sizeof( m ) );
	const float t = 1.0f / std::tan( fovyDeg * 3.14159265f / 360.0f );
	const float p[16] = { t / aspect, 0, 0, 0, 0, t, 0, 0, 0, 0, -1.0002f, -1, 0, 0, -0.20002f, 0 };
	std::memcpy( proj, p, sizeof( p ) );
}

struct Job
{
	int frame;
	uint8_t* pixels;
};

int main( int argc, char** argv )
{
	if( argc < 4 )
	{
		std::printf( "usage: rtcamp_batch scene.obj env.hdr outdir [--frame-range a b] [--frames N] [--size W H] [--res from to] [--steps K] [--png]\n" );
		std::printf( "  [instance 0] rtcamp_batch ... --frame-range 0 171\n  [instance 1] rtcamp_batch ... --frame-range 171 240\n" );
		return 0;
	}
	const char* objPath = argv[1];
	const char* hdrPath = argv[2];
	const std::string outDir = argv[3];
	int totalFrames = 240, beginFrame = 0, endFrame = -1, W = 1440, H = 900, fromRes = 256, toRes = 8192, steps = 8; // RTCamp.cpp:42-45,136-137,156
	bool png = false, dumpCameras = false;
	for( int i = 4; i < argc; i++ )
	{
		if( !std::strcmp( argv[i], "--dump-cameras" ) ) dumpCameras = true;
		else if( !std::strcmp( argv[i], "--frame-range" ) && i + 2 < argc ) { beginFrame = std::atoi( argv[i + 1] ); endFrame = std::atoi( argv[i + 2] ); i += 2; }
		else if( !std::strcmp( argv[i], "--frames" ) && i + 1 < argc ) totalFrames = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--size" ) && i + 2 < argc ) { W = std::atoi( argv[i + 1] ); H = std::atoi( argv[i + 2] ); i += 2; }
		else if( !std::strcmp( argv[i], "--res" ) && i + 2 < argc ) { fromRes = std::atoi( argv[i + 1] ); toRes = std::atoi( argv[i + 2] ); i += 2; }
		else if( !std::strcmp( argv[i], "--steps" ) && i + 1 < argc ) steps = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--png" ) ) png = true;
	}
	if( endFrame < 0 ) endFrame = totalFrames;

	std::vector<mvrt_io::V3> vertices, vcolors, vemissions;
	if( !mvrt_io::readObj( objPath, &vertices, &vcolors, &vemissions ) )
	{
		std::fprintf( stderr, "cannot read %s\n", objPath );
		return 1;
	}
	// a light: the top 6 % of the mesh glows (so m_hasEmission exercises the extra-sample branch)
	mvrt_io::V3 lo, dummy;
	float dps0;
	mvrt_io::boundingGrid( vertices, 1, &lo, &dps0 );
	for( size_t i = 0; i < vertices.size(); i++ )
		if( vertices[i].y > lo.y + 0.94f * dps0 ) vemissions[i] = mvrt_io::V3{ 1.0f, 0.85f, 0.6f };
	(void)dummy;

	void* stream = nullptr;
	mvrt::check( mvrt_stream_create( &stream ), "mvrt_stream_create" );
	char name[256];
	mvrt_device_name( name, sizeof( name ) );
	std::printf( "Device: %s\n", name );

	mvrt::PathTracer pt;
	pt.setup( stream );
	pt.resizeFrameBufferIfNeeded( stream, W, H );
	pt.loadHDRI( stream, hdrPath, hdrPath );

	// worker: image writer with a 4-buffer pool (RTCamp.cpp:126-130,174-193)
	std::mutex mu;
	std::condition_variable cv;
	std::deque<uint8_t*> pool;
	std::deque<Job> jobs;
	bool done = false;
	for( int i = 0; i < 4; i++ ) pool.push_back( new uint8_t[(size_t)W * H * 4] );
	std::thread writer( [&]() {
		for( ;; )
		{
			Job j;
			{
				std::unique_lock<std::mutex> lk( mu );
				cv.wait( lk, [&] { return done || !jobs.empty(); } );
				if( jobs.empty() ) return;
				j = jobs.front();
				jobs.pop_front();
			}
			char file[512];
			std::snprintf( file, sizeof( file ), "%s/%03d.%s", outDir.c_str(), j.frame, png ? "png" : "ppm" );
			if( png ) mvrt_io::writePngUncompressed( file, j.pixels, W, H );
			else mvrt_io::writePpm( file, j.pixels, W, H );
			{
				std::lock_guard<std::mutex> lk( mu );
				pool.push_back( j.pixels );
			}
			cv.notify_all();
		}
	} );

	const float boxWide = dps0; // largest extent of the mesh
	const float centre[3] = { lo.x + 0.5f * boxWide, lo.y + 0.5f * boxWide, lo.z + 0.5f * boxWide };
	for( int frame = beginFrame; frame < endFrame; frame++ )
	{
		// voxel size ramps linearly from boxWide/fromRes to boxWide/toRes over the whole animation (RTCamp.cpp:138-142)
		const float tt = (float)frame / (float)totalFrames;
		const float dps = ( boxWide / fromRes ) + ( ( boxWide / toRes ) - ( boxWide / fromRes ) ) * tt;
		const int resolution = (int)std::ceil( boxWide / dps );
		const int gridRes = (int)next_power_of_two( (uint32_t)resolution );
		pt.updateScene( vertices, vcolors, vemissions, stream, lo, dps, gridRes );

		const float ang = 6.2831853f * tt;
		const float eye[3] = { centre[0] + 1.6f * boxWide * std::cos( ang ), centre[1] + 0.5f * boxWide, centre[2] + 1.6f * boxWide * std::sin( ang ) };
		float view[16], proj[16];
		lookAt( eye, centre, 40.0f, (float)W / (float)H, view, proj );
		const float focus = 1.6f * boxWide, lensR = 0.004f * boxWide;
		if( dumpCameras ) // exact inputs of this frame (hex floats) so that a test can replay it through another binding
		{
			char file[512];
			std::snprintf( file, sizeof( file ), "%s/%03d.camera.txt", outDir.c_str(), frame );
			if( FILE* fp = std::fopen( file, "w" ) )
			{
				for( int k = 0; k < 16; k++ ) std::fprintf( fp, "%a ", view[k] );
				std::fprintf( fp, "\n" );
				for( int k = 0; k < 16; k++ ) std::fprintf( fp, "%a ", proj[k] );
				std::fprintf( fp, "\n%a %a %a %a %a %a %d\n", focus, lensR, lo.x, lo.y, lo.z, dps, gridRes );
				std::fclose( fp );
			}
		}

		pt.clearFrameBuffer( stream );
		for( int iteration = 0; iteration < steps; iteration++ ) pt.step( stream, view, proj, focus, lensR );

		uint8_t* buf = nullptr;
		{
			std::unique_lock<std::mutex> lk( mu );
			cv.wait( lk, [&] { return !pool.empty(); } );
			buf = pool.front();
			pool.pop_front();
		}
		pt.toImageAsync( stream, buf ); // resolve + DtoH (PathTracer.hpp:118-129)
		mvrt::check( mvrt_stream_synchronize( stream ), "sync" );
		{
			std::lock_guard<std::mutex> lk( mu );
			jobs.push_back( Job{ frame, buf } );
		}
		cv.notify_all();
		std::printf( "[frame %d] res( %d -> grid %d ) voxels %llu octree %.1f MB\n", frame, resolution, gridRes, (unsigned long long)pt.getNumberOfVoxels(), pt.getOctreeBytes() / 1e6 );
	}
	{
		std::lock_guard<std::mutex> lk( mu );
		done = true;
	}
	cv.notify_all();
	writer.join();
	for( uint8_t* p : pool ) delete[] p;
	pt.cleanUp();
	mvrt_stream_destroy( stream );
	std::printf( "done\n" );
	return 0;
}
