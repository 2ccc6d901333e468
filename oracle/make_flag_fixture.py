"""Dump the reference's command-line flags (name, type, default, choices) into tests/golden/reference_flags.json.

Build container only: imports /root/reference/src/args.py in place (it needs nothing but argparse/os/torch) and
introspects the parser that ``get_args`` builds, without running ``parse_args`` side effects (mkdir)."""
import argparse
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
import args as ref_args  # noqa: E402

parser = argparse.ArgumentParser("NLP GAN args")
ref_args.add_training_args(parser)
ref_args.add_data_args(parser)
ref_args.add_model_args(parser)
# the global flags are declared inline in get_args (args.py:208-256); capture them by intercepting parse_args
captured = {}
orig_parse = argparse.ArgumentParser.parse_args


def fake_parse(self, *a, **k):
    captured["parser"] = self
    raise SystemExit(0)


argparse.ArgumentParser.parse_args = fake_parse
try:
    ref_args.get_args()
except SystemExit:
    pass
argparse.ArgumentParser.parse_args = orig_parse
full = captured["parser"]
flags = {}
for act in full._actions:
    if not act.option_strings or act.dest == "help":
        continue
    flags[act.option_strings[0]] = {"dest": act.dest, "type": getattr(act.type, "__name__", str(act.type)),
                                    "default": act.default, "choices": list(act.choices) if act.choices else None}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_flags.json")
json.dump(flags, open(out, "w"), indent=1, sort_keys=True)
print(len(flags), "flags ->", out)
