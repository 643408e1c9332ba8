// Source-compatibility shim: the reference's <bvh/v2/platform.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
