// Build shim (ours): stands in for <windows.h> so that the reference's utils.cpp compiles with g++.
// Only what that translation unit names is provided; the one function is never called by the checker.
#pragma once
#define MAX_PATH 260
inline unsigned GetModuleFileNameA(void*, char* buf, unsigned n) { if (n) buf[0] = 0; return 0; }
