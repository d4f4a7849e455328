// TEST INFRASTRUCTURE: what `#include <hip/hip_runtime.h>` resolves to in the host emulation build (tests/cpp/hip_emu.hpp).
#pragma once
#include "../../hip_emu.hpp"
