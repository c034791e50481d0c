// tile_render.cpp -- the native multi-GPU host path of the north star: ONE process drives N GPUs of a node, the frame's 256-pixel blocks are
// dealt round-robin to the devices (mvrt_pt_set_tile), the read-only SVO / HDRI / PMJ tables are replicated per device (rebuilt
// deterministically from the same triangles), and after the K steps the per-device accumulation buffers are exchanged with ONE
// ncclAllGather (RCCL over xGMI) and assembled into the full frame on every device.  The reference has no multi-GPU path; its host
// structure (setup / loadHDRI / updateScene / N x step / toImage, voxPTGPU.cpp:88-193) is kept per device.
//
//   tile_render scene.obj env.hdr out.ppm [--gpus N] [--size W H] [--res R] [--steps K] [--frames F] [--dump-f32 file] [--dump-camera file]
//
// C++ on the header-only mirrors (include/mvrt/*.hpp) + the C-ABI (libmvrt_hip.so) + librccl.  Samples depend only on (global pixel,
// spp), so the assembled frame equals the 1-GPU frame bit for bit (tests/test_apps.py).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <functional>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "mvrt/PathTracer.hpp"
#include "scene_io.hpp"

#define NCCL_CHECK( expr )                                                                          \
	do                                                                                              \
	{                                                                                               \
		ncclResult_t r_ = ( expr );                                                                 \
		if( r_ != ncclSuccess )                                                                     \
		{                                                                                           \
			std::fprintf( stderr, "%s failed: %s\n", #expr, ncclGetErrorString( r_ ) );             \
			std::abort();                                                                           \
		}                                                                                           \
	} while( 0 )

// look-at view matrix + GL perspective, column-major (what prlib's GetCameraMatrix hands to PathTracer::step)
static void lookAt( const float eye[3], const float at[3], float fovyDeg, float aspect, float view[16], float proj[16] )
{
	float f[3] = { at[0] - eye[0], at[1] - eye[1], at[2] - eye[2] };
	const float fl = std::sqrt( f[0] * f[0] + f[1] * f[1] + f[2] * f[2] );
	for( float& c : f ) c /= fl;
	const float up[3] = { 0, 1, 0 };
	float s[3] = { f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0] };
	const float sl = std::sqrt( s[0] * s[0] + s[1] * s[1] + s[2] * s[2] );
	for( float& c : s ) c /= sl;
	const float u[3] = { s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0] };
	const float m[16] = { s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
						  -( s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2] ), -( u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2] ), f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2], 1 };
	std::memcpy( view, m, sizeof( m ) );
	const float t = 1.0f / std::tan( fovyDeg * 3.14159265f / 360.0f );
	const float p[16] = { t / aspect, 0, 0, 0, 0, t, 0, 0, 0, 0, -1.0002f, -1, 0, 0, -0.20002f, 0 };
	std::memcpy( proj, p, sizeof( p ) );
}

struct Device // everything one GPU owns
{
	int id = 0;
	void* stream = nullptr;
	std::unique_ptr<mvrt::PathTracer> pt;
	float* gathered = nullptr; // nDev x ownedPixels float4: every device's accumulation buffer after the all-gather
	float* frame = nullptr;	   // W x H float4, assembled
	uint8_t* frameU8 = nullptr;
	ncclComm_t comm = nullptr;
};

int main( int argc, char** argv )
{
	if( argc < 4 )
	{
		std::printf( "usage: tile_render scene.obj env.hdr out.ppm [--gpus N] [--size W H] [--res R] [--steps K] [--frames F] [--dump-f32 file] [--dump-camera file]\n" );
		return 0;
	}
	const char* objPath = argv[1];
	const char* hdrPath = argv[2];
	const char* outPath = argv[3];
	int wantGpus = 8, W = 1920, H = 1080, gridRes = 1024, steps = 4, frames = 1;
	const char *dumpF32 = nullptr, *dumpCamera = nullptr;
	for( int i = 4; i < argc; i++ )
	{
		if( !std::strcmp( argv[i], "--gpus" ) && i + 1 < argc ) wantGpus = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--size" ) && i + 2 < argc ) { W = std::atoi( argv[i + 1] ); H = std::atoi( argv[i + 2] ); i += 2; }
		else if( !std::strcmp( argv[i], "--res" ) && i + 1 < argc ) gridRes = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--steps" ) && i + 1 < argc ) steps = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--frames" ) && i + 1 < argc ) frames = std::atoi( argv[++i] );
		else if( !std::strcmp( argv[i], "--dump-f32" ) && i + 1 < argc ) dumpF32 = argv[++i];
		else if( !std::strcmp( argv[i], "--dump-camera" ) && i + 1 < argc ) dumpCamera = argv[++i];
	}
	int nAvail = 0;
	mvrt::check( mvrt_device_count( &nAvail ), "mvrt_device_count" );
	const int nDev = wantGpus < nAvail ? wantGpus : nAvail;
	if( nDev < 1 )
	{
		std::fprintf( stderr, "no GPU\n" );
		return 1;
	}

	std::vector<mvrt_io::V3> vertices, vcolors, vemissions;
	if( !mvrt_io::readObj( objPath, &vertices, &vcolors, &vemissions ) )
	{
		std::fprintf( stderr, "cannot read %s\n", objPath );
		return 1;
	}
	mvrt_io::V3 lo;
	float boxWide, dps;
	mvrt_io::boundingGrid( vertices, 1, &lo, &boxWide );
	mvrt_io::boundingGrid( vertices, gridRes, &lo, &dps );
	for( size_t i = 0; i < vertices.size(); i++ ) // a light: the top 6 % of the mesh glows (hasEmission: the extra-sample branch, voxKernel.cu:720-739)
		if( vertices[i].y > lo.y + 0.94f * boxWide ) vemissions[i] = mvrt_io::V3{ 1.0f, 0.85f, 0.6f };

	// ---- one communicator per device, all in this process ----
	std::vector<int> ids( nDev );
	for( int d = 0; d < nDev; d++ ) ids[d] = d;
	std::vector<ncclComm_t> comms( nDev );
	NCCL_CHECK( ncclCommInitAll( comms.data(), nDev, ids.data() ) );

	// ---- per-device setup: replicated scene, own tile ----
	std::vector<Device> dev( nDev );
	uint64_t owned = 0;
	const auto t0 = std::chrono::steady_clock::now();
	// one host thread per device: the replicated scene is built on all devices at once (N x setup in the time of one), and later every
	// device's frame is submitted by its own thread (a frame is ~120 asynchronous launches per device: serial submission would cost the
	// last device several milliseconds of a 10 ms frame)
	auto onEveryDevice = [&]( const std::function<void( int )>& body ) {
		std::vector<std::thread> th;
		for( int d = 0; d < nDev; d++ ) th.emplace_back( body, d );
		for( std::thread& t : th ) t.join();
	};
	onEveryDevice( [&]( int d ) {
		Device& D = dev[d];
		D.id = d;
		D.comm = comms[d];
		mvrt::check( mvrt_set_device( d ), "mvrt_set_device" );
		mvrt::check( mvrt_stream_create( &D.stream ), "mvrt_stream_create" );
		D.pt.reset( new mvrt::PathTracer() );
		D.pt->setup( D.stream );
		D.pt->setTile( d, nDev );
		D.pt->resizeFrameBufferIfNeeded( D.stream, W, H );
		D.pt->loadHDRI( D.stream, hdrPath, hdrPath );
		D.pt->updateScene( vertices, vcolors, vemissions, D.stream, lo, dps, gridRes );
		const uint64_t own = mvrt_pt_owned_pixels( D.pt->handle() ); // (the same on every device: padded to equal shares)
		mvrt::check( mvrt_malloc( (void**)&D.gathered, (uint64_t)nDev * own * 16 ), "mvrt_malloc" );
		mvrt::check( mvrt_malloc( (void**)&D.frame, (uint64_t)W * H * 16 ), "mvrt_malloc" );
		mvrt::check( mvrt_malloc( (void**)&D.frameU8, (uint64_t)W * H * 4 ), "mvrt_malloc" );
	} );
	owned = mvrt_pt_owned_pixels( dev[0].pt->handle() );
	const double setupS = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();

	const float centre[3] = { lo.x + 0.5f * boxWide, lo.y + 0.5f * boxWide, lo.z + 0.5f * boxWide };
	const float eye[3] = { centre[0] + 1.3f * boxWide, centre[1] + 0.6f * boxWide, centre[2] + 1.1f * boxWide };
	float view[16], proj[16];
	lookAt( eye, centre, 40.0f, (float)W / (float)H, view, proj );
	const float focus = 1.8f * boxWide, lensR = 0.004f * boxWide;
	if( dumpCamera )
	{
		if( FILE* fp = std::fopen( dumpCamera, "w" ) )
		{
			for( int k = 0; k < 16; k++ ) std::fprintf( fp, "%a ", view[k] );
			std::fprintf( fp, "\n" );
			for( int k = 0; k < 16; k++ ) std::fprintf( fp, "%a ", proj[k] );
			std::fprintf( fp, "\n%a %a %a %a %a %a %d\n", focus, lensR, lo.x, lo.y, lo.z, dps, gridRes );
			std::fclose( fp );
		}
	}

	// ---- frames: every device renders its tile share asynchronously, then one all-gather + assemble per device ----
	double bestMs = 1e30;
	for( int frame = 0; frame < frames; frame++ )
	{
		const auto f0 = std::chrono::steady_clock::now();
		onEveryDevice( [&]( int d ) {
			Device& D = dev[d];
			mvrt::check( mvrt_set_device( D.id ), "mvrt_set_device" );
			D.pt->clearFrameBuffer( D.stream );
			for( int k = 0; k < steps; k++ ) D.pt->step( D.stream, view, proj, focus, lensR );
			mvrt::check( mvrt_pt_join( D.pt->handle(), D.stream ), "mvrt_pt_join" ); // launches the deferred steps; D.stream waits for them
		} );
		NCCL_CHECK( ncclGroupStart() );
		for( Device& D : dev )
			NCCL_CHECK( ncclAllGather( mvrt_pt_framebuffer_dev( D.pt->handle() ), D.gathered, owned * 4, ncclFloat, D.comm, (hipStream_t)D.stream ) );
		NCCL_CHECK( ncclGroupEnd() );
		onEveryDevice( [&]( int d ) {
			Device& D = dev[d];
			mvrt::check( mvrt_set_device( D.id ), "mvrt_set_device" );
			mvrt::check( mvrt_pt_assemble_tiles( D.gathered, nDev, owned, W, H, D.frame, D.stream ), "mvrt_pt_assemble_tiles" );
			mvrt::check( mvrt_stream_synchronize( D.stream ), "sync" );
		} );
		const double ms = std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - f0 ).count();
		if( ms < bestMs ) bestMs = ms;
		std::printf( "[frame %d] %d GPU(s), %dx%d, %d spp: %.3f ms\n", frame, nDev, W, H, steps * 16, ms );
	}

	// ---- device 0 writes the image (renderResolve on the assembled frame) ----
	mvrt::check( mvrt_set_device( 0 ), "mvrt_set_device" );
	mvrt::check( mvrt_resolve_buffer( dev[0].frame, (uint64_t)W * H, dev[0].frameU8, dev[0].stream ), "mvrt_resolve_buffer" );
	std::vector<uint8_t> rgba( (size_t)W * H * 4 );
	mvrt::check( mvrt_memcpy_d2h( rgba.data(), dev[0].frameU8, rgba.size(), dev[0].stream ), "d2h" );
	if( !mvrt_io::writePpm( outPath, rgba.data(), W, H ) ) std::fprintf( stderr, "cannot write %s\n", outPath );
	if( dumpF32 )
	{
		std::vector<float> f( (size_t)W * H * 4 );
		// every device holds the same assembled frame: dump the LAST device's copy (so a test sees the gathered data of a non-zero rank too)
		mvrt::check( mvrt_set_device( dev[nDev - 1].id ), "mvrt_set_device" );
		mvrt::check( mvrt_memcpy_d2h( f.data(), dev[nDev - 1].frame, f.size() * 4, dev[nDev - 1].stream ), "d2h" );
		if( FILE* fp = std::fopen( dumpF32, "wb" ) )
		{
			std::fwrite( f.data(), 4, f.size(), fp );
			std::fclose( fp );
		}
	}
	std::printf( "devices %d, voxels %llu, setup %.2f s, best frame %.3f ms\n", nDev, (unsigned long long)dev[0].pt->getNumberOfVoxels(), setupS, bestMs );
	// rays of ONE frame, over all devices (intersect() calls: the same count the reference's megakernel would make)
	unsigned long long rays = 0;
	for( Device& D : dev )
	{
		mvrt::check( mvrt_set_device( D.id ), "mvrt_set_device" );
		mvrt_pt_stats st;
		mvrt::check( mvrt_pt_get_stats( D.pt->handle(), D.stream, &st ), "mvrt_pt_get_stats" );
		rays += st.rays / (unsigned long long)frames;
	}
	std::printf( "{\"app\": \"tile_render\", \"n_gpus\": %d, \"width\": %d, \"height\": %d, \"spp\": %d, \"grid_res\": %d, \"frames\": %d, \"best_frame_ms\": %.3f, \"rays_per_frame\": %llu, "
				 "\"mrays_per_s\": %.1f, \"setup_s\": %.2f, \"collective\": \"ncclAllGather of %llu float4 per device + assemble, inside the timed frame\"}\n",
				 nDev, W, H, steps * 16, gridRes, frames, bestMs, rays, (double)rays / bestMs / 1e3, setupS, (unsigned long long)owned );

	for( Device& D : dev )
	{
		mvrt::check( mvrt_set_device( D.id ), "mvrt_set_device" );
		D.pt->cleanUp();
		mvrt_free( D.gathered );
		mvrt_free( D.frame );
		mvrt_free( D.frameU8 );
		mvrt_stream_destroy( D.stream );
		ncclCommDestroy( D.comm );
	}
	return 0;
}
