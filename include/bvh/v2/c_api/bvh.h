/* Source-compatibility shim: the reference's C header <bvh/v2/c_api/bvh.h> maps onto include/bvh_amd.h, which re-declares
 * its types and functions with the same names, argument meaning and ownership rules (each citing the line it replaces). */
#ifndef BVH_V2_C_API_BVH_H
#define BVH_V2_C_API_BVH_H
#include "../../../bvh_amd.h"
#endif
