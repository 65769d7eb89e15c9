#!/usr/bin/env python3
"""Mutation fuzzing of the glTF loader under AddressSanitizer / UBSan (host code only, no GPU).

Builds tools/fuzz/load_scene.cpp with the host sources (`make -C tools/fuzz`), then writes N mutated copies of
tests/golden/cornell_box.gltf (one to three leaf values deleted or replaced by out-of-range, negative, huge or
wrongly typed values; the .bin now and then truncated) and loads each in a child process.  A load may fail with a
message; any other exit status (a sanitizer report, a signal) is printed and the offending file kept.

    python tools/fuzz/gltf_mutate.py [seed [count]]
"""
import copy
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tests", "golden", "cornell_box.gltf")
HERE = os.path.dirname(os.path.abspath(__file__))


def leaves(o, pre=()):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from leaves(v, pre + (k,))
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from leaves(v, pre + (i,))
    else:
        yield pre, o


def parent(d, path):
    for k in path[:-1]:
        d = d[k]
    return d


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    subprocess.run(["make", "-C", HERE, "load_scene"], check=True)
    doc = json.load(open(SRC))
    blob = open(SRC.replace(".gltf", ".bin"), "rb").read()
    paths = list(leaves(doc))
    rnd = random.Random(seed)
    work = tempfile.mkdtemp(prefix="gltf_fuzz_")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    bad = 0
    for it in range(count):
        d = copy.deepcopy(doc)
        what = []
        for _ in range(rnd.choice([1, 1, 2, 3])):
            p, v = rnd.choice(paths)
            try:
                if rnd.random() < 0.15:
                    q = parent(d, p)
                    q.pop(p[-1]) if isinstance(q, list) else q.__delitem__(p[-1])
                    what.append(("del", p))
                    continue
                if isinstance(v, bool) or not isinstance(v, (int, float)):
                    nv = rnd.choice([None, 5, -1, "x", [], {}, 1e30])
                else:
                    nv = rnd.choice([-1, 0, 1, 2, 7, 255, 65535, 2**31 - 1, 2**31, 2**32 - 1, 2**40, -2**31, 1e30, -1e30, 0.5,
                                     float(v) * 3, int(v) + 1, "7", None, [1], {}])
                parent(d, p)[p[-1]] = nv
                what.append((p, nv))
            except (KeyError, IndexError, TypeError):
                pass
        json.dump(d, open(os.path.join(work, "cornell_box.gltf"), "w"))
        b = blob[:rnd.randrange(0, len(blob))] if rnd.random() < 0.15 else blob
        open(os.path.join(work, "cornell_box.bin"), "wb").write(b)
        r = subprocess.run([os.path.join(HERE, "load_scene"), os.path.join(work, "cornell_box.gltf")], capture_output=True,
                           text=True, env=env, timeout=300)
        if r.returncode != 0:
            bad += 1
            keep = os.path.join(work, f"crash_{it}.gltf")
            shutil.copy(os.path.join(work, "cornell_box.gltf"), keep)
            print("exit status", r.returncode, what, keep, r.stderr[:400].replace("\n", " | "))
    print(f"{count} mutated files, {bad} ended in something other than a load or an error message ({work})")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
