// thread-local state of the TEST-ONLY host simulator (see hostsim.hpp)
#ifdef PXS_HOST_SIM
#include "hostsim.hpp"
namespace pxsim {
thread_local uint3_ t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx;
}
#endif
