// Source-compatibility shim: the reference's <bvh/v2/mini_tree_builder.h> maps onto the single-header mirror.
#pragma once
#include "bvh_amd.hpp"
