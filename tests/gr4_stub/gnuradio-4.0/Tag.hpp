// TEST-ONLY stand-in (see Block.hpp): gr::Tag lives in Block.hpp
#pragma once
#include "Block.hpp"
