// TEST-ONLY stand-in (see ../Block.hpp): apps/packet_receiver_file.cpp includes gnuradio4's Soapy block and uses nothing of it
#pragma once
