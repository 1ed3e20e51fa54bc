// MOCK: see rsworld.cuh in this directory
#pragma once
#include "rsworld.cuh"
