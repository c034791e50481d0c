// svo_build.hip -- placeholder until the GPU builder lands (next commit).
#include "launch.h"
int svoBuildFromTriangles( const float*, const float*, const float*, uint64_t, f3, float, int, hipStream_t, SvoBuildResult* )
{
	mvrtSetError( "mvrt_svo_build: GPU SVO construction is not implemented yet" );
	return 1;
}
