// concurrency_probe.hip -- can other kernels run while a persistent kernel that only spins (16 one-wave workgroups per CU, 8 KB of LDS and 80 VGPRs each) holds the GPU?
// hipcc --offload-arch=gfx950 -O3 tools/calib/concurrency_probe.hip -o /tmp/cprobe && /tmp/cprobe [lowprio 0|1] [wavesPerCU]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK( x ) do { hipError_t e = ( x ); if( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); exit( 1 ); } } while( 0 )
__global__ void __launch_bounds__( 64, 5 ) spin( volatile unsigned* stop, unsigned* alive, float* sink )
{
	__shared__ float lds[LDSF];
	lds[threadIdx.x] = threadIdx.x;
	float acc[48];
	for( int i = 0; i < 48; i++ ) acc[i] = lds[( threadIdx.x + i ) % LDSF];
	if( threadIdx.x == 0 ) atomicAdd( alive, 1u );
	unsigned long long t0 = wall_clock64();
	while( __hip_atomic_load( (unsigned*)stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) == 0u && wall_clock64() - t0 < 300000000ull )
	{
		__builtin_amdgcn_s_sleep( 127 );
		for( int i = 0; i < 48; i++ ) acc[i] = acc[i] * 1.0001f + 1.0f;
	}
	float s = 0;
	for( int i = 0; i < 48; i++ ) s += acc[i];
	if( s == 12345.f ) sink[0] = s;
}
__global__ void small( unsigned* c ) { if( threadIdx.x == 0 && blockIdx.x == 0 ) atomicAdd( c, 1u ); }
__global__ void __launch_bounds__( 1024 ) big( unsigned* c ) { __shared__ unsigned x[17]; x[threadIdx.x & 15] = 1; __syncthreads(); if( threadIdx.x == 0 ) atomicAdd( c, x[3] ); }
__global__ void __launch_bounds__( 256, 5 ) wide( unsigned* c, float* sink )
{
	float acc[70];
	for( int i = 0; i < 70; i++ ) acc[i] = threadIdx.x + i;
	for( int k = 0; k < 100; k++ ) for( int i = 0; i < 70; i++ ) acc[i] = acc[i] * 1.0001f + 1.0f;
	float s = 0; for( int i = 0; i < 70; i++ ) s += acc[i];
	if( s == 12345.f ) sink[0] = s;
	if( threadIdx.x == 0 && blockIdx.x == 0 ) atomicAdd( c, 1u );
}
int main( int argc, char** argv )
{
	const int lowprio = argc > 1 ? atoi( argv[1] ) : 1, wpc = argc > 2 ? atoi( argv[2] ) : 16;
	hipDeviceProp_t p; CK( hipGetDeviceProperties( &p, 0 ) );
	unsigned *stop, *alive, *cnt; float* sink;
	CK( hipMalloc( &stop, 4 ) ); CK( hipMalloc( &alive, 4 ) ); CK( hipMalloc( &cnt, 4 ) ); CK( hipMalloc( &sink, 4 ) );
	CK( hipMemset( stop, 0, 4 ) ); CK( hipMemset( alive, 0, 4 ) ); CK( hipMemset( cnt, 0, 4 ) );
	int lo, hi; CK( hipDeviceGetStreamPriorityRange( &lo, &hi ) );
	hipStream_t sp, s1, s2, side;
	if( lowprio ) CK( hipStreamCreateWithPriority( &sp, hipStreamNonBlocking, lo ) ); else CK( hipStreamCreateWithFlags( &sp, hipStreamNonBlocking ) );
	CK( hipStreamCreateWithFlags( &s1, hipStreamNonBlocking ) ); CK( hipStreamCreateWithFlags( &s2, hipStreamNonBlocking ) ); CK( hipStreamCreateWithFlags( &side, hipStreamNonBlocking ) );
	CK( hipDeviceSynchronize() );
	printf( "priority range %d..%d, low-priority pool stream %d, %d spinning waves per CU (%d CUs)\n", lo, hi, lowprio, wpc, p.multiProcessorCount );
	hipLaunchKernelGGL( spin, dim3( p.multiProcessorCount * wpc ), dim3( 64 ), 0, sp, stop, alive, sink );
	auto t0 = std::chrono::steady_clock::now();
	auto ms = [&]() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count(); };
	unsigned h = 0;
	for( int k = 0; k < 50 && h < (unsigned)( p.multiProcessorCount * wpc ); k++ ) { CK( hipMemcpyAsync( &h, alive, 4, hipMemcpyDeviceToHost, side ) ); CK( hipStreamSynchronize( side ) ); }
	printf( "%.2f ms: %u spinning waves alive\n", ms(), h );
	for( int k = 0; k < 20; k++ ) hipLaunchKernelGGL( small, dim3( 1 ), dim3( 64 ), 0, s1, cnt );
	CK( hipStreamSynchronize( s1 ) ); printf( "%.2f ms: 20 small kernels on stream 1 done\n", ms() );
	for( int k = 0; k < 20; k++ ) hipLaunchKernelGGL( big, dim3( 1 ), dim3( 1024 ), 0, s2, cnt );
	CK( hipStreamSynchronize( s2 ) ); printf( "%.2f ms: 20 1024-thread kernels on stream 2 done\n", ms() );
	for( int k = 0; k < 20; k++ ) hipLaunchKernelGGL( wide, dim3( 2048 ), dim3( 256 ), 0, s1, cnt, sink );
	CK( hipStreamSynchronize( s1 ) ); printf( "%.2f ms: 20 grids of 2048 x 256 threads (80 VGPRs) on stream 1 done\n", ms() );
	hipLaunchKernelGGL( small, dim3( 1 ), dim3( 64 ), 0, 0, cnt );
	CK( hipStreamSynchronize( 0 ) ); printf( "%.2f ms: a kernel on the null stream done\n", ms() );
	unsigned one = 1; CK( hipMemcpyAsync( stop, &one, 4, hipMemcpyHostToDevice, side ) ); CK( hipStreamSynchronize( side ) );
	CK( hipDeviceSynchronize() ); printf( "%.2f ms: spinner stopped, all done\n", ms() );
	return 0;
}
