// TEST-ONLY stand-in (see ../Block.hpp) for gnuradio4's Soapy source as far as the reference's flowgraph sources use it:
// apps/packet_receiver_soapy.cpp:33-37 emplaces gr::blocks::soapy::SoapyBlock<c64, 1UZ> with the settings "device",
// "sample_rate", "rx_center_frequency", "rx_gains" and connects its port "out" (:63) to the receiver's detector;
// apps/packet_receiver_file.cpp includes the header and uses nothing of it.  No device is opened and nothing is
// produced: the stand-in's scheduler executes no flowgraph.
#pragma once
#include <gnuradio-4.0/Block.hpp>
#include <gnuradio-4.0/reflection.hpp>

#include <string>
#include <vector>

namespace gr::blocks::soapy {
template <typename T, std::size_t nPorts = 1UZ>
class SoapyBlock : public gr::Block<SoapyBlock<T, nPorts>>
{
    static_assert(nPorts == 1UZ, "the stand-in declares the single-port form the reference's apps use");

public:
    gr::PortOut<T> out;
    std::string device;
    std::string device_parameter;
    float sample_rate = 1'000'000.f;
    std::vector<double> rx_center_frequency{ 107'000'000. };
    std::vector<double> rx_bandwdith{ 500'000. };
    std::vector<double> rx_gains{ 5. };
};
} // namespace gr::blocks::soapy

ENABLE_REFLECTION_FOR_TEMPLATE_FULL((typename T, std::size_t nPorts), (gr::blocks::soapy::SoapyBlock<T, nPorts>), out, device,
                                    device_parameter, sample_rate, rx_center_frequency, rx_bandwdith, rx_gains);
