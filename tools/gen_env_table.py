"""Regenerates the two lists that describe the library's environment switches -- `enum env_id` in vexcl_amd/csrc/common.hpp and `kEnvNames`
in vexcl_amd/csrc/runtime.hip -- from the ENV_<NAME> tokens used in vexcl_amd/csrc (run after adding an env(ENV_...) site)."""
import glob, os, re
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = set()
for f in glob.glob(os.path.join(root, "vexcl_amd/csrc/*.hip")) + glob.glob(os.path.join(root, "vexcl_amd/csrc/*.hpp")):
    names |= set(re.findall(r"\bENV_((?:VEXHIP|VEXCL)_[A-Z0-9_]+)\b", open(f).read()))
names = sorted(names)
p = os.path.join(root, "vexcl_amd/csrc/common.hpp"); s = open(p).read()
a = s.index("enum env_id {"); b = s.index("};", a) + 3
s = s[:a] + "enum env_id {\n" + "".join("    ENV_%s,\n" % n for n in names) + "    ENV_COUNT\n};\n" + s[b:]
open(p, "w").write(s)
p = os.path.join(root, "vexcl_amd/csrc/runtime.hip"); s = open(p).read()
a = s.index("const char *const kEnvNames[ENV_COUNT] = {"); b = s.index("};", a) + 3
s = s[:a] + "const char *const kEnvNames[ENV_COUNT] = {\n" + "".join('    "%s",\n' % n for n in names) + "};\n" + s[b:]
open(p, "w").write(s)
print(len(names), "switches")
