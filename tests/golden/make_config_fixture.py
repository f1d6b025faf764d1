#!/usr/bin/env python3
"""Generates tests/golden/config_defaults.json: the configuration keys, C++ types and defaults of the reference's drivers,
derived THE REFERENCE'S OWN WAY and stored as data (key -> default, key -> type, key order).

* stage 5 (5-sim-genome): the C preprocessor over `src/prepare/defaults.py.in`, which includes `../config_entries.inc` under
  `#define X(var, type, value) #var: value,` -- exactly the rule of `src/prepare/Makefile:1-2,14-17` (`$(CPP) -xc -std=c99 -P`);
  the produced module is executed and its DEFAULT_CONFIG recorded (prepare/run.py:21-23 merges the user's JSON over it).  A second
  pass with X defined to yield the type column records the C++ types (simulation_common/simulation_config.cc reads every key
  with that type, all mandatory).
* stage 4 (4-sim-ab box / sphere): the X-macro table `X_CONFIG_JSON_PARAMETERS` of `simulation_config.hpp` expanded by the same
  preprocessor with X defined to yield (type, name, default).

Only inputs and outputs are stored; run here (the reference tree does not exist on the GPU box):
    python tests/golden/make_config_fixture.py"""
import json
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
CPP = ["cpp", "-xc", "-std=c99", "-P"]      # prepare/Makefile:1 (CPPFLAGS = -xc -std=c99 -P)


def stage5():
    src = os.path.join(REF, "5-sim-genome/src/prepare/defaults.py.in")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "defaults.py")
        subprocess.check_call(CPP + ["-o", out, src])      # `$(CPP) $(CPPFLAGS) -o $@ $<`
        ns = {}
        exec(compile(open(out).read(), out, "exec"), ns)
        defaults = ns["DEFAULT_CONFIG"]
        # the type column: the same include under another X
        probe = os.path.join(tmp, "types.c")
        with open(probe, "w") as f:
            f.write('#define X(var, type, value) @var@type@\n#define V(...) [__VA_ARGS__]\n'
                    f'#include "{os.path.join(REF, "5-sim-genome/src/config_entries.inc")}"\n')
        text = subprocess.check_output(CPP + [probe], text=True)
    types = dict(re.findall(r"@\s*(\w+)\s*@\s*([\w:]+)\s*@", text))
    assert list(types) == list(defaults), "type pass and default pass disagree on the keys"
    return {"source": "5-sim-genome/src/config_entries.inc via src/prepare/defaults.py.in (prepare/Makefile:14-17)",
            "keys": list(defaults), "defaults": dict(defaults), "ctypes": types}


def stage4(rel):
    text = open(os.path.join(REF, rel)).read()
    m = re.search(r"#define X_CONFIG_JSON_PARAMETERS(?:[^\n]*\\\n)*[^\n]*\n", text)
    assert m, rel
    with tempfile.TemporaryDirectory() as tmp:
        probe = os.path.join(tmp, "ab.c")
        with open(probe, "w") as f:
            f.write(m.group(0) + "#define X(T, var, init) @T@var@init@\nX_CONFIG_JSON_PARAMETERS\n")
        out = subprocess.check_output(CPP + [probe], text=True)
    rows = re.findall(r"@\s*([\w:]+)\s*@\s*(\w+)\s*@\s*([^@]*?)\s*@", out)
    defaults, types = {}, {}
    for T, var, init in rows:
        types[var] = T
        defaults[var] = json.loads(init) if T == "std::string" else (float(init) if T == "md::scalar" else int(init))
    return {"source": rel, "keys": [r[1] for r in rows], "defaults": defaults, "ctypes": types}


def main():
    fx = {"stage5": stage5(),
          "stage4_box": stage4("4-sim-ab/box/src/simulation/simulation_config.hpp"),
          "stage4_sphere": stage4("4-sim-ab/sphere/src/simulation_config.hpp")}
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump(fx, f, indent=1)
        f.write("\n")
    print({k: len(v["keys"]) for k, v in fx.items()})


if __name__ == "__main__":
    main()
