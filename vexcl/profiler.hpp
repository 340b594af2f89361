#ifndef VEXCL_PROFILER_HPP
#define VEXCL_PROFILER_HPP
// vex::stopwatch and vex::profiler (reference: vexcl/profiler.hpp:92-353):
// named, nestable wall-clock intervals; tic_cl fences every queue first.
#include <chrono>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "backend.hpp"

namespace vex {

template <class Clock = std::chrono::high_resolution_clock>
class stopwatch {
    public:
        stopwatch() : tlap(0), ttot(0), n(0) { tic(); }
        void tic() { start = Clock::now(); }
        double toc() {
            double d = std::chrono::duration<double>(Clock::now() - start).count();
            tlap = d; ttot += d; ++n;
            return d;
        }
        double average() const { return n ? ttot / n : 0; }
        double total() const { return ttot; }
        size_t tics() const { return n; }
    private:
        typename Clock::time_point start;
        double tlap, ttot;
        size_t n;
};

template <class Clock = std::chrono::high_resolution_clock>
class profiler {
    public:
        profiler(const std::vector<backend::command_queue> &queue = std::vector<backend::command_queue>(),
                 const std::string &name = "Profile")
            : queue(queue), root(std::make_shared<unit>(name)) { stack.push_back(root.get()); root->watch.tic(); }

        void tic_cpu(const std::string &name) { enter(name); }
        void tic_cl(const std::string &name) {
            for (const auto &q : queue) q.finish();
            enter(name);
            stack.back()->cl = true;             // toc() waits for the queues before reading the clock (profiler.hpp:249-269)
        }
        double toc(const std::string & /*name*/ = "") {
            precondition(stack.size() > 1, "profiler::toc() without tic");
            unit *u = stack.back();
            if (u->cl) for (const auto &q : queue) q.finish();
            double d = u->watch.toc();
            stack.pop_back();
            return d;
        }
        void reset() { root->children.clear(); stack.assign(1, root.get()); root->watch = stopwatch<Clock>(); }

        void print(std::ostream &out) {
            root->watch.toc();
            double total = root->watch.total();
            out << std::endl;
            root->print(out, 0, total);
        }
    private:
        struct unit {
            std::string name; bool cl = false; stopwatch<Clock> watch;
            std::vector<std::shared_ptr<unit>> children;
            explicit unit(const std::string &n) : name(n) {}
            void print(std::ostream &out, unsigned level, double total) const {
                out << std::string(2 * level, ' ') << name << ": " << std::fixed << std::setprecision(6)
                    << watch.total() << " sec.";
                if (total > 0) out << " (" << std::setprecision(2) << 100 * watch.total() / total << "%)";
                if (watch.tics() > 1) out << " [" << watch.tics() << "x; avg " << std::setprecision(1)
                                          << watch.average() * 1e6 << " usec.]";
                out << std::endl;
                for (const auto &c : children) c->print(out, level + 1, total);
            }
        };
        void enter(const std::string &name) {
            unit *p = stack.back();
            for (auto &c : p->children) if (c->name == name) { c->watch.tic(); stack.push_back(c.get()); return; }
            p->children.push_back(std::make_shared<unit>(name));
            p->children.back()->watch = stopwatch<Clock>();
            stack.push_back(p->children.back().get());
        }
        std::vector<backend::command_queue> queue;
        std::shared_ptr<unit> root;
        std::vector<unit *> stack;
};

template <class Clock>
std::ostream &operator<<(std::ostream &os, profiler<Clock> &prof) { prof.print(os); return os; }

} // namespace vex
#endif
