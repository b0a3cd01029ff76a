// TEST-ONLY (see Block.hpp in this directory)
#pragma once
#include "Block.hpp"
