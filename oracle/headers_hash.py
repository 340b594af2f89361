"""sha256 over vexcl/**/*.hpp and include/vexhip.h (sorted by path): what oracle/_ref was built from."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def headers_hash():
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "vexhip.h")]
    for base, _, names in os.walk(os.path.join(ROOT, "vexcl")):
        files += [os.path.join(base, n) for n in names if n.endswith(".hpp")]
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(headers_hash())
