// the CPU stand-in has one header for the whole runtime
#pragma once
#include "hip_runtime.h"
