// TEST-ONLY stand-in (see ../gnuradio-4.0/Block.hpp, which holds the pmtv slice the blocks use)
#pragma once
#include <gnuradio-4.0/Block.hpp>
