// TEST-ONLY stand-in (see Block.hpp) for gr::scheduler::Simple<policy>{std::move(fg)}.runAndWait()
// (benchmarks/benchmark_syncword_detection.cpp:87-93).  It is not a runtime: runAndWait() checks the graph the
// flowgraph source built (every stream input has its upstream) and calls start() / stop() of every block -- for the
// drop-in blocks that is the C ABI's create call on the GPU -- then reports that it cannot execute a flowgraph.
// Set GR4_STUB_LIFECYCLE=0 to skip start() / stop() (machines without a GPU: the drop-ins have no CPU fallback).
#pragma once
#include <gnuradio-4.0/Graph.hpp>

#include <cstdlib>

namespace gr::scheduler {
enum class ExecutionPolicy { singleThreaded, multiThreaded };

struct Error {
    std::string message;
};
// std::expected<void, Error> as far as the flowgraph sources use it
struct RunResult {
    std::optional<Error> err;
    bool has_value() const { return !err.has_value(); }
    const std::string& error() const { return err->message; }
};

template <ExecutionPolicy Policy = ExecutionPolicy::singleThreaded>
class Simple
{
    gr::Graph _graph;

public:
    explicit Simple(gr::Graph&& g) : _graph(std::move(g)) {}
    const gr::Graph& graph() const { return _graph; }
    RunResult runAndWait()
    {
        const char* lc = std::getenv("GR4_STUB_LIFECYCLE");
        const bool lifecycle = !(lc && lc[0] == '0');
        try {
            if (lifecycle) {
                for (auto& start : _graph.starters) start();
                for (auto& stop : _graph.stoppers) stop();
            }
        } catch (const std::exception& e) {
            return { Error{ e.what() } };
        }
        return { Error{ "gr4 stand-in: " + std::to_string(_graph.blocks.size()) + " blocks, " +
                        std::to_string(_graph.edges.size()) + " edges, lifecycle " + (lifecycle ? "run" : "skipped") + ", " +
                        std::to_string(_graph.needed_a_device) + " settings calls deferred for want of a device" +
                        "; the test stand-in does not execute flowgraphs" } };
    }
};
} // namespace gr::scheduler
