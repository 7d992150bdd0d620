// (see Headers/Common.hpp: the shim declares everything src/main.cpp names)
#pragma once
#include "Common.hpp"
