// Stand-in for <boost/program_options.hpp>, as far as the reference's examples use it (examples/benchmark.cpp,
// examples/fft_benchmark.cpp): options_description / add_options()(name, [value<T>(&var)->default_value(v),] help),
// parse_command_line, store, notify, variables_map::count, printing the description.  Options are written
// --name value, --name=value or -x value; bool values accept 1/0/true/false/on/off.  Test infrastructure only.
#ifndef VEX_REF_SHIM_BOOST_PROGRAM_OPTIONS_HPP
#define VEX_REF_SHIM_BOOST_PROGRAM_OPTIONS_HPP
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace boost { namespace program_options {

struct value_semantic {
    std::function<void(const std::string &)> assign;     // parse and store
    std::function<void()> apply_default;
    bool has_default = false, is_bool = false;
    bool is_switch = false;                              // bool_switch: takes no argument, presence means true
    std::function<void()> apply_implicit;                // implicit_value: used when the option comes without a value
    std::string default_text;
};

template <class T>
struct typed_value {
    std::shared_ptr<value_semantic> sem;
    T *target;
    explicit typed_value(T *t) : sem(new value_semantic), target(t) {
        T *p = t;
        sem->is_bool = std::is_same<T, bool>::value;
        sem->assign = [p](const std::string &s) { if (p) parse(s, *p); };
    }
    typed_value *default_value(const T &v) {
        T *p = target;
        sem->has_default = true;
        std::ostringstream o; o << v; sem->default_text = o.str();
        sem->apply_default = [p, v]() { if (p) *p = v; };
        return this;
    }
    typed_value *implicit_value(const T &v) {
        T *p = target;
        sem->apply_implicit = [p, v]() { if (p) *p = v; };
        return this;
    }
    static void parse(const std::string &s, bool &out) {
        if (s == "1" || s == "true" || s == "on" || s == "yes") out = true;
        else if (s == "0" || s == "false" || s == "off" || s == "no") out = false;
        else throw std::runtime_error("bad boolean option value: " + s);
    }
    template <class U> static void parse(const std::string &s, U &out) {
        std::istringstream i(s);
        if (!(i >> out)) throw std::runtime_error("bad option value: " + s);
    }
    static void parse(const std::string &s, std::string &out) { out = s; }
};
template <class T> typed_value<T> *value(T *target = nullptr) { return new typed_value<T>(target); }   // lives as long as the program, like Boost's
inline typed_value<bool> *bool_switch(bool *target = nullptr) {
    typed_value<bool> *v = new typed_value<bool>(target);
    v->default_value(false);
    v->sem->is_switch = true;
    v->sem->apply_implicit = [target]() { if (target) *target = true; };
    return v;
}

struct option {
    std::string long_name, short_name, help;
    std::shared_ptr<value_semantic> sem;     // null: a flag
};

class options_description {
    public:
        explicit options_description(const std::string &caption = "") : caption(caption) {}
        struct adder {
            options_description &d;
            adder &operator()(const char *name, const char *help) { d.add(name, nullptr, help); return *this; }
            template <class T> adder &operator()(const char *name, typed_value<T> *v, const char *help = "") { d.add(name, v->sem, help); return *this; }
        };
        adder add_options() { return adder{*this}; }
        const option *find(const std::string &n, bool is_short) const {
            for (const auto &o : opts) if ((is_short ? o.short_name : o.long_name) == n) return &o;
            return nullptr;
        }
        std::vector<option> opts;
        std::string caption;
    private:
        void add(const std::string &name, std::shared_ptr<value_semantic> sem, const std::string &help) {
            option o; o.help = help; o.sem = sem;
            const size_t comma = name.find(',');
            o.long_name = name.substr(0, comma);
            if (comma != std::string::npos) o.short_name = name.substr(comma + 1);
            opts.push_back(o);
        }
};
inline std::ostream &operator<<(std::ostream &os, const options_description &d) {
    os << d.caption << ":\n";
    for (const auto &o : d.opts) {
        os << "  ";
        if (!o.short_name.empty()) os << "-" << o.short_name << " [ --" << o.long_name << " ]"; else os << "--" << o.long_name;
        if (o.sem) { os << " arg"; if (o.sem->has_default) os << " (=" << o.sem->default_text << ")"; }
        os << "  " << o.help << "\n";
    }
    return os;
}

struct parsed_options {
    const options_description *desc;
    std::vector<std::pair<const option *, std::string>> found;
};
inline parsed_options parse_command_line(int argc, const char *const *argv, const options_description &desc) {
    parsed_options p; p.desc = &desc;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], val; bool has_val = false;
        const option *o = nullptr;
        if (a.rfind("--", 0) == 0) {
            const size_t eq = a.find('=');
            if (eq != std::string::npos) { val = a.substr(eq + 1); has_val = true; a = a.substr(0, eq); }
            o = desc.find(a.substr(2), false);
        } else if (a.size() >= 2 && a[0] == '-') {
            o = desc.find(a.substr(1, 1), true);
            if (a.size() > 2) { val = a.substr(2); has_val = true; }
        }
        if (!o) throw std::runtime_error("unrecognised option '" + a + "'");
        bool implicit = false;
        if (o->sem && !has_val) {
            const bool next_is_value = i + 1 < argc && !(argv[i + 1][0] == '-' && argv[i + 1][1] != '\0' && !(argv[i + 1][1] >= '0' && argv[i + 1][1] <= '9'));
            if (o->sem->is_switch || (o->sem->apply_implicit && !next_is_value)) implicit = true;
            else {
                if (i + 1 >= argc) throw std::runtime_error("the required argument for option '" + a + "' is missing");
                val = argv[++i];
            }
        }
        p.found.push_back(std::make_pair(o, implicit ? std::string("\x01implicit") : val));
    }
    return p;
}
template <class C> parsed_options parse_command_line(int argc, C **argv, const options_description &desc) {
    return parse_command_line(argc, const_cast<const char *const *>(argv), desc);
}

class variables_map {
    public:
        size_t count(const std::string &name) const { auto i = seen.find(name); return i == seen.end() ? 0 : i->second; }
        std::map<std::string, size_t> seen;
        std::vector<std::function<void()>> pending;
};
inline void store(const parsed_options &p, variables_map &vm) {
    for (const auto &o : p.desc->opts)
        if (o.sem && o.sem->has_default) { vm.pending.push_back(o.sem->apply_default); ++vm.seen[o.long_name]; }
    for (const auto &f : p.found) {
        ++vm.seen[f.first->long_name];
        if (f.first->sem) {
            auto sem = f.first->sem; std::string v = f.second;
            if (v == "\x01implicit") vm.pending.push_back(sem->apply_implicit);
            else vm.pending.push_back([sem, v]() { sem->assign(v); });
        }
    }
}
inline void notify(variables_map &vm) { for (auto &f : vm.pending) f(); vm.pending.clear(); }

} }
#endif
