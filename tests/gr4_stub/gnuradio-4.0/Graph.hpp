// TEST-ONLY stand-in (see Block.hpp) for the slice of gnuradio4's Graph.hpp that the reference's flowgraph sources
// use: fg.emplaceBlock<T>({settings}), fg.connect<"out">(a).to<"in">(b) != gr::ConnectionResult::SUCCESS
// (benchmarks/benchmark_syncword_detection.cpp:29-85, packet_receiver.hpp:76-265, apps/packet_receiver_file.cpp:31-70).
// connect() resolves both port names at COMPILE time against the blocks' ENABLE_REFLECTION lists and checks that the
// two ports carry the same item type: a drop-in class with a missing / renamed / retyped port does not compile.
#pragma once
#include <gnuradio-4.0/Block.hpp>

namespace gr {
// (ConnectionResult: Block.hpp, where PortOut::resizeBuffer() returns it too)

class Graph : public stub::Graph
{
public:
    struct Edge {
        const void *src, *dst;
        std::string src_port, dst_port;
    };
    std::vector<Edge> edges;

    template <meta::fixed_string SrcName, typename Src>
    struct Connector {
        Graph& g;
        Src& src;
        template <meta::fixed_string DstName, typename Dst>
        [[nodiscard]] ConnectionResult to(Dst& dst)
        {
            auto& out = stub::Reflect<Src>::template member<SrcName>(src);
            auto& in = stub::Reflect<Dst>::template member<DstName>(dst);
            using Out = std::remove_cvref_t<decltype(out)>;
            using In = std::remove_cvref_t<decltype(in)>;
            static_assert(stub::PortLike<Out> && !Out::is_input, "connect<name>(block): not an output port");
            static_assert(stub::PortLike<In> && In::is_input, ".to<name>(block): not an input port");
            static_assert(std::is_same_v<typename Out::value_type, typename In::value_type>,
                          "the two ports of an edge carry different item types");
            if (!g.owns(&src) || !g.owns(&dst)) return ConnectionResult::FAILED;
            if (!in.links.empty()) return ConnectionResult::FAILED; // an input port has one upstream
            const std::string sn(SrcName.data), dn(DstName.data);
            out.links.push_back({ &dst, dn });
            in.links.push_back({ &src, sn });
            g.edges.push_back({ &src, &dst, sn, dn });
            return ConnectionResult::SUCCESS;
        }
    };
    template <meta::fixed_string SrcName, typename Src>
    Connector<SrcName, Src> connect(Src& src)
    {
        return { *this, src };
    }
    // fg.connect(a, "out"s, b, "in#2"s) (packet_transmitter_pdu.hpp:273-401): ports by run-time name, elements of a
    // std::vector of ports as "name#k"; the same checks as the compile-time form, made when the graph is built
    template <typename Src, typename Dst>
    [[nodiscard]] ConnectionResult connect(Src& src, const std::string& src_port, Dst& dst, const std::string& dst_port)
    {
        stub::PortRef out, in;
        if (!stub::Reflect<Src>::port(src, src_port, out) || !stub::Reflect<Dst>::port(dst, dst_port, in))
            return ConnectionResult::FAILED; // no such port
        if (out.is_input || !in.is_input || *out.item != *in.item) return ConnectionResult::FAILED;
        if (!owns(&src) || !owns(&dst) || !in.links->empty()) return ConnectionResult::FAILED;
        out.links->push_back({ &dst, dst_port });
        in.links->push_back({ &src, src_port });
        edges.push_back({ &src, &dst, src_port, dst_port });
        return ConnectionResult::SUCCESS;
    }
    bool owns(const void* b) const
    {
        for (const auto& p : blocks)
            if (p.get() == b) return true;
        return false;
    }
};
} // namespace gr
