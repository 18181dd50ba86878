// Build shim (ours): glUtils.cpp includes this header with a lower-case initial (case-insensitive file system upstream).
#pragma once
#include "utils/ShaderRegistry.hpp"
