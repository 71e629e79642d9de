#pragma once
#include <stdint.h>
namespace zke { struct DfaDev { uint32_t dummy; }; }
