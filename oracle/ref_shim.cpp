/*
 * ref_shim.cpp -- C entry points over the two pieces of the REFERENCE that compile unmodified in
 * this image: morton.hpp (morton.hpp:5-113) and libs/smhasher/MurmurHash3.cpp.  This file contains
 * no reference code; it includes the reference headers from $(REF) at build time (oracle/Makefile,
 * target `ref`) and the resulting library lives only in oracle/_ref/ (git-ignored).
 * TEST INFRASTRUCTURE: used by tests/test_oracle_pins.py to pin the oracle's integer helpers the
 * same way the reference's own unittest.cpp:106-132,183-227 does.
 */
#include <stdint.h>
#include "morton.hpp"
#include "MurmurHash3.h"

extern "C" {
uint64_t ref_morton_naive( uint32_t x, uint32_t y, uint32_t z ) { return encode2mortonCode_Naive( x, y, z ); }
uint64_t ref_morton_pdep( uint32_t x, uint32_t y, uint32_t z ) { return encode2mortonCode_PDEP( x, y, z ); }
uint64_t ref_morton_magicbits( uint32_t x, uint32_t y, uint32_t z ) { return encode2mortonCode_magicbits( x, y, z ); }
void ref_morton_decode_naive( uint64_t m, uint32_t* xyz ) { decodeMortonCode_Naive( m, xyz, xyz + 1, xyz + 2 ); }
void ref_morton_decode_pext( uint64_t m, uint32_t* xyz ) { decodeMortonCode_PEXT( m, xyz, xyz + 1, xyz + 2 ); }
void ref_morton_decode_magicbits( uint64_t m, uint32_t* xyz ) { decodeMortonCode_magicBits( m, xyz, xyz + 1, xyz + 2 ); }
void ref_morton_batch( const uint32_t* xyz, int64_t n, uint64_t* naive, uint64_t* pdep, uint64_t* magic )
{
	for( int64_t i = 0; i < n; i++ )
	{
		naive[i] = encode2mortonCode_Naive( xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2] );
		pdep[i] = encode2mortonCode_PDEP( xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2] );
		magic[i] = encode2mortonCode_magicbits( xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2] );
	}
}
uint32_t ref_murmur3_x86_32( const uint32_t* words, int nWords, uint32_t seed )
{
	uint32_t h;
	MurmurHash3_x86_32( words, nWords * 4, seed, &h );
	return h;
}
}
