// gather_rate.hip -- how many divergent dword loads per second a gfx950 CU sustains (the traversal's child-pointer fetch is one).
//   footprint: 16 KiB (L1 resident), 2 MiB (L2 resident), 128 MiB (Infinity Cache), 8 GiB and 128 GiB (HBM; the larger one also
//              exceeds the TLB reach, like the 122 GB node array of the config-5 stress octree)
//   pattern  : every lane a random 64-byte line ("div"), or all 64 lanes of a wave-instruction in one line ("same")
//   mode     : 8 independent loads in flight per lane (throughput), or one dependent chain per lane (latency bound, what a
//              traversal lane does), at 7 waves per SIMD
// prints G lane-loads/s for the chip, and cycles per wave-instruction per CU at 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/gather_rate.hip -o build/gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint32_t hash( uint32_t x )
{
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}
template <int SAME>
__global__ void __launch_bounds__( 64 ) kIndep( const uint32_t* __restrict__ p, uint32_t lineMask, int iters, uint32_t* out )
{
	uint32_t acc = 0, s = ( blockIdx.x * 64 + ( SAME ? 0 : threadIdx.x ) ) * 2654435761u + 1u;
	for( int i = 0; i < iters; i++ )
	{
		uint32_t v[8];
#pragma unroll
		for( int k = 0; k < 8; k++ )
		{
			s = hash( s + k );
			v[k] = p[(uint64_t)( s & lineMask ) * 16 + ( threadIdx.x & 15 )];
		}
#pragma unroll
		for( int k = 0; k < 8; k++ ) acc += v[k];
	}
	out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__( 64 ) kChase( const uint32_t* __restrict__ p, uint32_t lineMask, int iters, uint32_t* out )
{
	uint32_t s = ( blockIdx.x * 64 + threadIdx.x ) * 2654435761u + 1u;
	for( int i = 0; i < iters; i++ )
	{
		const uint32_t v = p[(uint64_t)( s & lineMask ) * 16 + ( threadIdx.x & 15 )]; // the buffer holds random words: next address depends on the load
		s = hash( s ^ v );
	}
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ void kFill( uint32_t* p, uint64_t n )
{
	for( uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x ) p[i] = hash( (uint32_t)i );
}
int main()
{
	hipDeviceProp_t pr;
	(void)hipGetDeviceProperties( &pr, 0 );
	const int cus = pr.multiProcessorCount, grid = cus * 28;
	uint32_t *buf, *out;
	const uint64_t maxBytes = 128ull << 30;
	(void)hipMalloc( (void**)&buf, maxBytes );
	(void)hipMalloc( (void**)&out, (size_t)grid * 64 * 4 );
	hipLaunchKernelGGL( kFill, dim3( 4096 ), dim3( 256 ), 0, 0, buf, maxBytes / 4 );
	(void)hipDeviceSynchronize();
	hipEvent_t e0, e1;
	(void)hipEventCreate( &e0 );
	(void)hipEventCreate( &e1 );
	const uint64_t sizes[5] = { 16ull << 10, 2ull << 20, 128ull << 20, 8ull << 30, 128ull << 30 };
	const char* names[5] = { "16 KiB (L1)", "2 MiB (L2)", "128 MiB (MALL)", "8 GiB (HBM)", "128 GiB (HBM)" };
	for( int si = 0; si < 5; si++ )
	{
		const uint32_t lineMask = (uint32_t)( sizes[si] / 64 - 1 );
		for( int mode = 0; mode < 3; mode++ )
		{
			const int iters = mode == 2 ? 2000 : 250;
			float ms = 0;
			for( int rep = 0; rep < 2; rep++ )
			{
				(void)hipEventRecord( e0 );
				if( mode == 0 ) hipLaunchKernelGGL( kIndep<0>, dim3( grid ), dim3( 64 ), 0, 0, buf, lineMask, iters, out );
				if( mode == 1 ) hipLaunchKernelGGL( kIndep<1>, dim3( grid ), dim3( 64 ), 0, 0, buf, lineMask, iters, out );
				if( mode == 2 ) hipLaunchKernelGGL( kChase, dim3( grid ), dim3( 64 ), 0, 0, buf, lineMask, iters, out );
				(void)hipEventRecord( e1 );
				(void)hipEventSynchronize( e1 );
				(void)hipEventElapsedTime( &ms, e0, e1 );
			}
			const double loads = (double)grid * 64 * iters * ( mode == 2 ? 1 : 8 );
			const double perCuInstr = ms * 1e-3 * 2.4e9 / ( loads / 64 / cus );
			printf( "%-14s %-28s %8.1f G lane-loads/s  %7.1f cycles per wave-instruction per CU\n", names[si],
					mode == 0 ? "divergent, 8 in flight/lane" : ( mode == 1 ? "same line, 8 in flight/lane" : "divergent, dependent chain" ), loads / ms / 1e6, perCuInstr );
			fflush( stdout );
		}
	}
	return 0;
}
