// Source-compatibility shim: the reference's <bvh/v2/thread_pool.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
