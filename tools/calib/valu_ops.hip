// valu_ops.hip -- issue cost of the VALU opcodes the traversal step is made of (gfx950, 7 waves per SIMD, no memory).
// Each kernel runs a long unrolled stream of ONE opcode on 8 independent register chains; prints SIMD-cycles per
// wave64 instruction at 2.4 GHz.  build: hipcc --offload-arch=gfx950 -O3 tools/calib/valu_ops.hip -o build/valu_ops
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2000
#define REP8( X ) X( 0 ) X( 1 ) X( 2 ) X( 3 ) X( 4 ) X( 5 ) X( 6 ) X( 7 )
#define KERNEL( NAME, ASM )                                                                                                   \
	__global__ void __launch_bounds__( 64 ) NAME( uint32_t* out, uint32_t seed )                                             \
	{                                                                                                                         \
		uint32_t r[8], q[8];                                                                                                  \
		unsigned long long sm = 0x5555555555555555ull ^ seed;                                                                \
		for( int k = 0; k < 8; k++ )                                                                                          \
		{                                                                                                                     \
			r[k] = threadIdx.x * 17u + seed + k;                                                                              \
			q[k] = r[k] ^ 0x3f800000u;                                                                                        \
		}                                                                                                                     \
		for( int i = 0; i < ITERS; i++ )                                                                                      \
		{                                                                                                                     \
			_Pragma( "unroll" ) for( int u = 0; u < 4; u++ )                                                                  \
			{                                                                                                                 \
				asm volatile( ASM( 0 ) ASM( 1 ) ASM( 2 ) ASM( 3 ) ASM( 4 ) ASM( 5 ) ASM( 6 ) ASM( 7 )                             \
							  : "+v"( r[0] ), "+v"( r[1] ), "+v"( r[2] ), "+v"( r[3] ), "+v"( r[4] ), "+v"( r[5] ), "+v"( r[6] ), "+v"( r[7] ) \
							  : "v"( q[0] ), "v"( q[1] ), "v"( q[2] ), "v"( q[3] ), "v"( q[4] ), "v"( q[5] ), "v"( q[6] ), "v"( q[7] ), "s"( sm ) \
							  : "vcc" );                                                                                      \
			}                                                                                                                 \
		}                                                                                                                     \
		uint32_t a = 0;                                                                                                       \
		for( int k = 0; k < 8; k++ ) a += r[k];                                                                               \
		out[blockIdx.x * 64 + threadIdx.x] = a;                                                                               \
	}
// operand numbering: %0-%7 = r (in/out), %8-%15 = q (in), %16 = sgpr pair
#define A_ADD( k ) "v_add_u32 %" #k ", %" #k ", %8\n\t"
#define A_MULF( k ) "v_mul_f32 %" #k ", %" #k ", %9\n\t"
#define A_FMA( k ) "v_fma_f32 %" #k ", %" #k ", %9, %10\n\t"
#define A_MAX3( k ) "v_max3_f32 %" #k ", %" #k ", %9, %10\n\t"
#define A_BFI( k ) "v_bfi_b32 %" #k ", %" #k ", %9, %10\n\t"
#define A_BFE( k ) "v_bfe_i32 %" #k ", %" #k ", %9, 1\n\t"
#define A_ANDOR( k ) "v_and_or_b32 %" #k ", %" #k ", %9, %10\n\t"
#define A_CNDS( k ) "v_cndmask_b32 %" #k ", %" #k ", %9, %16\n\t"
#define A_CMPS( k ) "v_cmp_lt_f32 vcc, %" #k ", %9\n\t"
#define A_LSHR( k ) "v_lshrrev_b32 %" #k ", 1, %" #k "\n\t"
KERNEL( kAdd, A_ADD )
KERNEL( kMulF, A_MULF )
KERNEL( kFma, A_FMA )
KERNEL( kMax3, A_MAX3 )
KERNEL( kBfi, A_BFI )
KERNEL( kBfe, A_BFE )
KERNEL( kAndOr, A_ANDOR )
KERNEL( kCndS, A_CNDS )
KERNEL( kCmp, A_CMPS )
KERNEL( kLshr, A_LSHR )
// packed fp32 and 64-bit shifts need register pairs
__global__ void __launch_bounds__( 64 ) kPkMul( uint32_t* out, uint32_t seed )
{
	typedef float v2f __attribute__( ( ext_vector_type( 2 ) ) );
	v2f r[8], q = { 1.0000001f, 0.9999999f };
	for( int k = 0; k < 8; k++ ) r[k] = v2f{ (float)( threadIdx.x + k + seed ), 1.5f };
	for( int i = 0; i < ITERS; i++ )
	{
#pragma unroll
		for( int u = 0; u < 4; u++ )
			asm volatile( "v_pk_mul_f32 %0, %0, %8\n\tv_pk_mul_f32 %1, %1, %8\n\tv_pk_mul_f32 %2, %2, %8\n\tv_pk_mul_f32 %3, %3, %8\n\t"
						  "v_pk_mul_f32 %4, %4, %8\n\tv_pk_mul_f32 %5, %5, %8\n\tv_pk_mul_f32 %6, %6, %8\n\tv_pk_mul_f32 %7, %7, %8"
						  : "+v"( r[0] ), "+v"( r[1] ), "+v"( r[2] ), "+v"( r[3] ), "+v"( r[4] ), "+v"( r[5] ), "+v"( r[6] ), "+v"( r[7] )
						  : "v"( q ) );
	}
	float a = 0;
	for( int k = 0; k < 8; k++ ) a += r[k].x + r[k].y;
	out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)a;
}
__global__ void __launch_bounds__( 64 ) kShl64( uint32_t* out, uint32_t seed )
{
	uint64_t r[8];
	for( int k = 0; k < 8; k++ ) r[k] = (uint64_t)( threadIdx.x + k + seed ) * 0x9E3779B97F4A7C15ull;
	for( int i = 0; i < ITERS; i++ )
	{
#pragma unroll
		for( int u = 0; u < 4; u++ )
			asm volatile( "v_lshlrev_b64 %0, 3, %0\n\tv_lshlrev_b64 %1, 3, %1\n\tv_lshlrev_b64 %2, 3, %2\n\tv_lshlrev_b64 %3, 3, %3\n\t"
						  "v_lshlrev_b64 %4, 3, %4\n\tv_lshlrev_b64 %5, 3, %5\n\tv_lshlrev_b64 %6, 3, %6\n\tv_lshlrev_b64 %7, 3, %7"
						  : "+v"( r[0] ), "+v"( r[1] ), "+v"( r[2] ), "+v"( r[3] ), "+v"( r[4] ), "+v"( r[5] ), "+v"( r[6] ), "+v"( r[7] ) );
	}
	uint64_t a = 0;
	for( int k = 0; k < 8; k++ ) a += r[k];
	out[blockIdx.x * 64 + threadIdx.x] = (uint32_t)a;
}
template <class K>
void run( K kern, const char* name, uint32_t* out, int cus )
{
	hipEvent_t e0, e1;
	(void)hipEventCreate( &e0 );
	(void)hipEventCreate( &e1 );
	const int grid = cus * 4 * 7;
	hipLaunchKernelGGL( kern, dim3( grid ), dim3( 64 ), 0, 0, out, 1u );
	(void)hipEventRecord( e0 );
	hipLaunchKernelGGL( kern, dim3( grid ), dim3( 64 ), 0, 0, out, 2u );
	(void)hipEventRecord( e1 );
	(void)hipEventSynchronize( e1 );
	float ms;
	(void)hipEventElapsedTime( &ms, e0, e1 );
	const double instrPerSimd = 7.0 * ITERS * 32;
	printf( "%-16s %.2f SIMD-cycles per wave64 instruction at 2.4 GHz (%.3f ms)\n", name, ms * 1e-3 * 2.4e9 / instrPerSimd, ms );
	fflush( stdout );
}
int main()
{
	hipDeviceProp_t p;
	(void)hipGetDeviceProperties( &p, 0 );
	uint32_t* out;
	(void)hipMalloc( (void**)&out, (size_t)p.multiProcessorCount * 28 * 64 * 4 );
	const int c = p.multiProcessorCount;
	run( kAdd, "v_add_u32", out, c );
	run( kLshr, "v_lshrrev_b32", out, c );
	run( kMulF, "v_mul_f32", out, c );
	run( kFma, "v_fma_f32", out, c );
	run( kPkMul, "v_pk_mul_f32", out, c );
	run( kMax3, "v_max3_f32", out, c );
	run( kBfi, "v_bfi_b32", out, c );
	run( kBfe, "v_bfe_i32", out, c );
	run( kAndOr, "v_and_or_b32", out, c );
	run( kCndS, "v_cndmask (sgpr)", out, c );
	run( kCmp, "v_cmp_lt_f32", out, c );
	run( kShl64, "v_lshlrev_b64", out, c );
	return 0;
}
