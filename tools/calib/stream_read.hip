// stream_read.hip -- the streaming-read ceiling of the box ("also report against a measured streaming-read ceiling"): every lane
// reads consecutive 16-byte words (one wave-instruction = 1 KiB contiguous), grid-stride over a buffer far larger than the 256 MB Infinity
// Cache, several passes; prints GB/s per buffer size.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/stream_read.hip -o build/stream_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK( x ) do { hipError_t e = ( x ); if( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); return 1; } } while( 0 )
__global__ void __launch_bounds__( 256 ) kRead( const uint4* __restrict__ p, uint64_t n, uint32_t* out )
{
	uint32_t acc = 0;
	const uint64_t stride = (uint64_t)gridDim.x * 256 * 4;
	for( uint64_t i = (uint64_t)blockIdx.x * 256 * 4 + threadIdx.x; i + 3 * 256 < n; i += stride ) // four independent loads in flight per lane
	{
		const uint4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
		acc += a.x ^ a.w ^ b.y ^ b.w ^ c.x ^ c.z ^ d.y ^ d.z;
	}
	if( acc == 0x12345678u ) out[0] = acc; // keep the loads
}
int main()
{
	hipDeviceProp_t prop;
	CHECK( hipGetDeviceProperties( &prop, 0 ) );
	uint32_t* out;
	CHECK( hipMalloc( &out, 4 ) );
	for( uint64_t gib : { 1ull, 8ull, 32ull } )
	{
		const uint64_t bytes = gib << 30, n = bytes / 16;
		uint4* p;
		CHECK( hipMalloc( &p, bytes ) );
		CHECK( hipMemset( p, 1, bytes ) );
		hipEvent_t e0, e1;
		CHECK( hipEventCreate( &e0 ) );
		CHECK( hipEventCreate( &e1 ) );
		const int grid = prop.multiProcessorCount * 8, passes = 10;
		hipLaunchKernelGGL( kRead, dim3( grid ), dim3( 256 ), 0, 0, p, n, out ); // warm-up (page tables)
		CHECK( hipEventRecord( e0 ) );
		for( int k = 0; k < passes; k++ ) hipLaunchKernelGGL( kRead, dim3( grid ), dim3( 256 ), 0, 0, p, n, out );
		CHECK( hipEventRecord( e1 ) );
		CHECK( hipEventSynchronize( e1 ) );
		float ms = 0;
		CHECK( hipEventElapsedTime( &ms, e0, e1 ) );
		printf( "stream read %2llu GiB x %d passes: %.1f GB/s\n", (unsigned long long)gib, passes, (double)bytes * passes / ( ms * 1e6 ) );
		CHECK( hipFree( p ) );
	}
	return 0;
}
