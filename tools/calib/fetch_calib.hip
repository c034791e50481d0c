// fetch_calib.hip -- calibrate rocprofv3 FETCH_SIZE for the two read patterns that matter here (guide:
// MI355X_MICROARCH.md "HBM": FETCH_SIZE halves wide coalesced streams; other widths must be calibrated).
//   kStream : 16 B per lane coalesced read of 2 GiB  (known bytes = 2 GiB)
//   kGather : one dword per lane from a random 64-byte line of an 8 GiB buffer, 2^26 accesses
//             (working set >> 256 MiB Infinity Cache; known: 2^26 distinct-line touches)
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o build/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void kStream( const float4* __restrict__ p, uint64_t n, float* out )
{
	float acc = 0;
	for( uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x )
	{
		float4 v = p[i];
		acc += v.x + v.y + v.z + v.w;
	}
	if( acc == 12345.678f ) out[0] = acc;
}
// the path tracer's shade / generate kernels: one dword per lane, consecutive lanes consecutive addresses (SoA arrays): 256 B per wave-load
__global__ void kStream4( const float* __restrict__ p, uint64_t n, float* out )
{
	float acc = 0;
	for( uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x ) acc += p[i];
	if( acc == 12345.678f ) out[0] = acc;
}
// ... and the matching store
__global__ void kStore4( float* __restrict__ p, uint64_t n )
{
	for( uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x ) p[i] = (float)i;
}
__global__ void kGather( const uint32_t* __restrict__ p, uint64_t nLines, uint64_t nAccess, float* out )
{
	uint32_t acc = 0;
	for( uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nAccess; i += (uint64_t)gridDim.x * blockDim.x )
	{
		uint64_t h = i * 0x9E3779B97F4A7C15ull;
		h ^= h >> 29;
		h *= 0xBF58476D1CE4E5B9ull;
		h ^= h >> 32;
		uint64_t line = h % nLines;
		acc += p[line * 16 + ( i & 15 )];
	}
	if( acc == 0xDEADBEEF ) out[0] = (float)acc;
}
int main()
{
	const uint64_t bytes = 8ull << 30;
	void* buf;
	float* out;
	hipMalloc( &buf, bytes );
	hipMalloc( (void**)&out, 4 );
	hipMemset( buf, 1, bytes );
	hipDeviceSynchronize();
	hipLaunchKernelGGL( kStream, dim3( 4096 ), dim3( 256 ), 0, 0, (const float4*)buf, ( 2ull << 30 ) / 16, out );
	hipDeviceSynchronize();
	hipLaunchKernelGGL( kGather, dim3( 4096 ), dim3( 256 ), 0, 0, (const uint32_t*)buf, bytes / 64, 1ull << 26, out );
	hipDeviceSynchronize();
	hipLaunchKernelGGL( kStream4, dim3( 4096 ), dim3( 256 ), 0, 0, (const float*)buf, ( 2ull << 30 ) / 4, out );
	hipDeviceSynchronize();
	hipLaunchKernelGGL( kStore4, dim3( 4096 ), dim3( 256 ), 0, 0, (float*)buf, ( 2ull << 30 ) / 4 );
	hipDeviceSynchronize();
	printf( "stream bytes %llu ; gather accesses %llu ; 4-byte-per-lane stream read and store: %llu bytes each\n", 2ull << 30, 1ull << 26, 2ull << 30 );
	return 0;
}
