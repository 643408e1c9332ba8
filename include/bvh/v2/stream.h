// Source-compatibility shim: the reference's <bvh/v2/stream.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
