// Build shim (ours): the reference includes this MSVC-only header; nothing from it is needed on Linux.
#pragma once
