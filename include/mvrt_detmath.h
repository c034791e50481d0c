/*
 * mvrt_detmath.h -- deterministic fp32 transcendental functions shared by the
 * HIP kernels and the CPU oracle.
 *
 * Why this exists: the reference renderer calls fast device intrinsics on the
 * GPU (INTRIN_COS/SIN/POW = __cosf/__sinf/__expf(y*__logf(x)),
 * /root/reference/vectorMath.hpp:86-92) and libm on the host (:93-97), so the
 * reference's own CPU and GPU builds already differ in the last bits of every
 * Lambert direction.  To make "HIP path == CPU oracle" a BIT-EXACT statement we
 * pin the five transcendental functions the hot path uses to one definition
 * that is built only from IEEE-754 +,-,*,/ and integer ops (no fma, no libm),
 * compiled with -ffp-contract=off on both sides.  sqrt and division are IEEE
 * correctly rounded on both sides and are used directly.
 *
 * Accuracy (checked in tests/test_detmath.py against libm): sin/cos <= 2 ulp on
 * [-4pi, 4pi], atan2 <= 3 ulp, pow(x, 1/2.2) <= 8 ulp on (0, 64].
 *
 * Polynomials are the classic Cephes single-precision minimax sets.
 */
#ifndef MVRT_DETMATH_H
#define MVRT_DETMATH_H

#include <stdint.h>

#if defined( __HIPCC__ )
#define MVRT_HD __host__ __device__ inline
#else
#define MVRT_HD inline
#endif

#define MVRT_PI 3.14159265358979323846264338327950288f /* == reference PI, vectorMath.hpp:78 */

MVRT_HD uint32_t mvrt_f2u( float f )
{
	union { float f; uint32_t u; } c; c.f = f; return c.u;
}
MVRT_HD float mvrt_u2f( uint32_t u )
{
	union { float f; uint32_t u; } c; c.u = u; return c.f;
}

/* floor for |x| < 2^23, exact, no libm: truncate then fix negatives */
MVRT_HD float mvrt_floor_small( float x )
{
	float t = (float)(int)x;
	return ( t > x ) ? t - 1.0f : t;
}

/* sin and cos of x (|x| < ~1e4) */
MVRT_HD void mvrt_sincos( float x, float* s, float* c )
{
	/* k = nearest integer to x * 2/pi */
	float kf = mvrt_floor_small( x * 0.636619772367581343f + 0.5f );
	int k = (int)kf;
	/* Cody-Waite, pi/2 split in three; kf*DP1 is exact (DP1 has 8 significant bits) */
	float r = x - kf * 1.5703125f;
	r = r - kf * 4.837512969970703125e-4f;
	r = r - kf * 7.54978995489188216e-8f;
	float z = r * r;
	float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
	ps = ps * z - 1.6666654611e-1f;
	ps = ps * z * r + r;
	float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
	pc = pc * z + 4.166664568298827e-2f;
	pc = pc * z * z - 0.5f * z + 1.0f;
	switch( k & 3 )
	{
	case 0: *s = ps; *c = pc; break;
	case 1: *s = pc; *c = -ps; break;
	case 2: *s = -ps; *c = -pc; break;
	default: *s = -pc; *c = ps; break;
	}
}
MVRT_HD float mvrt_sin( float x ) { float s, c; mvrt_sincos( x, &s, &c ); return s; }
MVRT_HD float mvrt_cos( float x ) { float s, c; mvrt_sincos( x, &s, &c ); return c; }

/* atan(a) for a >= 0 */
MVRT_HD float mvrt_atan_pos( float a )
{
	float y0;
	if( a > 2.414213562373095f )
	{
		y0 = 1.5707963267948966f;
		a = -( 1.0f / a );
	}
	else if( a > 0.4142135623730950f )
	{
		y0 = 0.7853981633974483f;
		a = ( a - 1.0f ) / ( a + 1.0f );
	}
	else
	{
		y0 = 0.0f;
	}
	float z = a * a;
	float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
	p = p * z + 1.99777106478e-1f;
	p = p * z - 3.33329491539e-1f;
	p = p * z * a + a;
	return y0 + p;
}

/* atan2(y, x), result in (-pi, pi]; atan2(0,0) = 0 */
MVRT_HD float mvrt_atan2( float y, float x )
{
	float ay = y < 0.0f ? -y : y;
	float ax = x < 0.0f ? -x : x;
	float r;
	if( ax == 0.0f )
	{
		r = ( ay == 0.0f ) ? 0.0f : 1.5707963267948966f;
	}
	else
	{
		r = mvrt_atan_pos( ay / ax );
	}
	if( x < 0.0f )
	{
		r = MVRT_PI - r;
	}
	return y < 0.0f ? -r : r;
}

/* natural log for finite x > 0 (denormals treated through scaling) */
MVRT_HD float mvrt_log( float x )
{
	int e = 0;
	uint32_t u = mvrt_f2u( x );
	if( u < 0x00800000u ) /* denormal: scale by 2^23 */
	{
		x = x * 8388608.0f;
		u = mvrt_f2u( x );
		e = -23;
	}
	e += (int)( u >> 23 ) - 126;
	float m = mvrt_u2f( ( u & 0x007FFFFFu ) | 0x3F000000u ); /* [0.5, 1) */
	if( m < 0.707106781186547524f )
	{
		e = e - 1;
		m = m + m - 1.0f;
	}
	else
	{
		m = m - 1.0f;
	}
	float z = m * m;
	float p = 7.0376836292e-2f * m - 1.1514610310e-1f;
	p = p * m + 1.1676998740e-1f;
	p = p * m - 1.2420140846e-1f;
	p = p * m + 1.4249322787e-1f;
	p = p * m - 1.6668057665e-1f;
	p = p * m + 2.0000714765e-1f;
	p = p * m - 2.4999993993e-1f;
	p = p * m + 3.3333331174e-1f;
	p = p * m * z;
	float fe = (float)e;
	p = p + fe * -2.12194440e-4f;
	p = p - 0.5f * z;
	float r = m + p;
	r = r + fe * 0.693359375f;
	return r;
}

/* exp(x) for x in [-87, 88] (clamped outside) */
MVRT_HD float mvrt_exp( float x )
{
	if( x > 88.0f ) x = 88.0f;
	if( x < -87.0f ) x = -87.0f;
	float nf = mvrt_floor_small( x * 1.44269504088896341f + 0.5f );
	int n = (int)nf;
	x = x - nf * 0.693359375f;
	x = x - nf * -2.12194440e-4f;
	float z = x * x;
	float p = 1.9875691500e-4f * x + 1.3981999507e-3f;
	p = p * x + 8.3334519073e-3f;
	p = p * x + 4.1665795894e-2f;
	p = p * x + 1.6666665459e-1f;
	p = p * x + 5.0000001201e-1f;
	p = p * z + x + 1.0f;
	return p * mvrt_u2f( (uint32_t)( n + 127 ) << 23 );
}

/* pow(x, y) for x >= 0 as exp(y*log(x)); pow(0, y>0) = 0.  Mirrors the shape of
 * the reference's device INTRIN_POW (vectorMath.hpp:90). */
MVRT_HD float mvrt_pow( float x, float y )
{
	if( !( x > 0.0f ) )
	{
		return 0.0f;
	}
	return mvrt_exp( y * mvrt_log( x ) );
}

#endif /* MVRT_DETMATH_H */
