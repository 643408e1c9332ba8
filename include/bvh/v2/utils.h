// Source-compatibility shim: the reference's <bvh/v2/utils.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
