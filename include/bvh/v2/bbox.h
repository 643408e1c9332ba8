// Source-compatibility shim: the reference's <bvh/v2/bbox.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
