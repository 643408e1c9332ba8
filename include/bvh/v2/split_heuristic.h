// Source-compatibility shim: the reference's <bvh/v2/split_heuristic.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
