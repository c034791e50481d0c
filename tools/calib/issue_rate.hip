// issue_rate.hip -- what a gfx950 SIMD sustains for the traversal kernel's INSTRUCTION MIX, without any memory access.
//   mode 0: VALU only (independent v_bfi / v_add chains)           -> VALU instr/s per SIMD = clock / 4 when VALU bound
//   mode 1: VALU + SALU interleaved at the traversal loop's ratio (112 : 60)
//   mode 2: mode 1 + conditional branches over small VALU groups (12 per 112 VALU), all lanes active
//   mode 3: mode 2 with a lane-dependent predicate (divergent: both sides execute via exec masking)
// One wave per workgroup, grid = CUs * 4 * wavesPerSimd.  Prints instr/s per SIMD for each mode and occupancy.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/issue_rate.hip -o build/issue_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4000
template <int MODE>
__global__ void __launch_bounds__( 64 ) k( uint32_t* out, uint32_t seed )
{
	uint32_t a = threadIdx.x + seed, b = a * 3u, c = a ^ 0x55u, d = a + 7u, e = b + 1u, f = c + 2u, g = d + 3u, h = e + 5u;
	uint32_t s0 = seed, s1 = seed + 1u;
	for( int i = 0; i < ITERS; i++ )
	{
		// 16 VALU per group, 7 groups = 112 VALU per iteration
#pragma unroll
		for( int grp = 0; grp < 7; grp++ )
		{
			if( MODE >= 2 )
			{
				// a branch every ~9 VALU; predicate uniform (mode 2) or lane dependent (mode 3)
				const bool p = MODE == 2 ? ( ( s0 + grp ) & 1u ) != 0u : ( ( a >> grp ) & 1u ) != 0u;
				if( p )
				{
					asm volatile( "v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0" : "+v"( a ), "+v"( b ), "+v"( c ), "+v"( d ) );
				}
				else
				{
					asm volatile( "v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0" : "+v"( e ), "+v"( f ), "+v"( g ), "+v"( h ) );
				}
				asm volatile( "v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t"
							  "v_add_u32 %4, %4, %5\n\tv_xor_b32 %5, %5, %6\n\tv_add_u32 %6, %6, %7\n\tv_xor_b32 %7, %7, %4\n\t"
							  "v_add_u32 %0, %0, %5\n\tv_xor_b32 %1, %1, %6\n\tv_add_u32 %2, %2, %7\n\tv_xor_b32 %3, %3, %4"
							  : "+v"( a ), "+v"( b ), "+v"( c ), "+v"( d ), "+v"( e ), "+v"( f ), "+v"( g ), "+v"( h ) );
			}
			else
			{
				asm volatile( "v_add_u32 %0, %0, %1\n\tv_xor_b32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_xor_b32 %3, %3, %0\n\t"
							  "v_add_u32 %4, %4, %5\n\tv_xor_b32 %5, %5, %6\n\tv_add_u32 %6, %6, %7\n\tv_xor_b32 %7, %7, %4\n\t"
							  "v_add_u32 %0, %0, %5\n\tv_xor_b32 %1, %1, %6\n\tv_add_u32 %2, %2, %7\n\tv_xor_b32 %3, %3, %4\n\t"
							  "v_add_u32 %4, %4, %1\n\tv_xor_b32 %5, %5, %2\n\tv_add_u32 %6, %6, %3\n\tv_xor_b32 %7, %7, %0"
							  : "+v"( a ), "+v"( b ), "+v"( c ), "+v"( d ), "+v"( e ), "+v"( f ), "+v"( g ), "+v"( h ) );
			}
			if( MODE >= 1 )
			{
				// ~8.5 SALU per group -> 60 per iteration
				asm volatile( "s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\t"
							  "s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %0"
							  : "+s"( s0 ), "+s"( s1 )::"scc" );
				if( grp & 1 ) asm volatile( "s_add_u32 %0, %0, %1" : "+s"( s0 ), "+s"( s1 )::"scc" );
			}
		}
	}
	out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h + s0 + s1;
}
template <int MODE>
void run( uint32_t* out, int cus, int wps, const char* what )
{
	hipEvent_t e0, e1;
	(void)hipEventCreate( &e0 );
	(void)hipEventCreate( &e1 );
	const int grid = cus * 4 * wps;
	hipLaunchKernelGGL( k<MODE>, dim3( grid ), dim3( 64 ), 0, 0, out, 1u );
	hipEventRecord( e0 );
	hipLaunchKernelGGL( k<MODE>, dim3( grid ), dim3( 64 ), 0, 0, out, 2u );
	hipEventRecord( e1 );
	hipEventSynchronize( e1 );
	float ms;
	hipEventElapsedTime( &ms, e0, e1 );
	const double valu = 112.0 * ITERS * wps; // per SIMD
	printf( "mode %d (%s) waves/SIMD %d: %.3f ms -> %.1f M VALU instr/s per SIMD (x4 cycles = %.2f GHz-equivalent), %.3f us per wave-iteration\n", MODE, what, wps, ms,
			valu / ms / 1e3, valu / ms / 1e3 * 4 / 1e3, ms * 1e3 / ITERS );
}
int main()
{
	hipDeviceProp_t p;
	hipGetDeviceProperties( &p, 0 );
	uint32_t* out;
	hipMalloc( (void**)&out, (size_t)p.multiProcessorCount * 4 * 8 * 64 * 4 );
	for( int wps : { 1, 2, 4, 7 } )
	{
		run<0>( out, p.multiProcessorCount, wps, "VALU only" );
		run<1>( out, p.multiProcessorCount, wps, "VALU+SALU" );
		run<2>( out, p.multiProcessorCount, wps, "VALU+SALU+uniform branches" );
		run<3>( out, p.multiProcessorCount, wps, "VALU+SALU+divergent branches" );
	}
	return 0;
}
