// Build shim (ours), force-included: MSVC lets the reference call isnan() unqualified.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>
using std::isnan;
