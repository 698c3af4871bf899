// Single translation unit of libzigp.so: the dense and Kronecker paths share the GEMM core and kernels of
// zigp_kernels.h (plain __global__ definitions), so they are compiled together.
#include "zigp_dense.hip"
#include "zigp_kron.hip"
#include "zigp_comm.hip"
