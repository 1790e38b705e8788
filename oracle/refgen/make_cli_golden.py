"""Container-only generator (reads /root/reference, which never travels): the reference's command-line surface as DATA.

  tests/golden/train_cli.json = {
     "flags":   {flag name: {"default": ..., "action": ..., "type": ...}}  from the add_argument calls of train.py:40-112 (ast, no import:
                train.py imports deepspeed / peft / tensorboard, which are not installed),
     "launch_lines": {script name: [argv after `train.py`]}                 from train_scripts/*.sh,
     "infer_imports": [names infer_iground.py:26-27 imports from `train`] }

tests/test_train_cli_host.py checks grove_amd.train.parse_args against it.  Run:  python oracle/refgen/make_cli_golden.py
"""
import ast
import glob
import json
import os
import shlex

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "train_cli.json")


def flags_of(path, fn_name="parse_args"):
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == fn_name)
    out = {}
    for call in ast.walk(fn):
        if isinstance(call, ast.Call) and isinstance(call.func, ast.Attribute) and call.func.attr == "add_argument":
            name = ast.literal_eval(call.args[0])
            kw = {}
            for k in call.keywords:
                if k.arg in ("default", "action", "choices"):
                    kw[k.arg] = ast.literal_eval(k.value)
                elif k.arg == "type":
                    kw["type"] = ast.unparse(k.value)
            out[name] = kw
    return out


def launch_lines():
    out = {}
    for sh in sorted(glob.glob(os.path.join(REF, "train_scripts", "*.sh"))):
        for line in open(sh):
            if "train.py" in line and line.strip().startswith("$LAUNCHER"):
                argv = shlex.split(line.split("train.py", 1)[1].split(">", 1)[0])
                out[os.path.basename(sh)] = argv
    return out


def infer_imports():
    tree = ast.parse(open(os.path.join(REF, "infer_iground.py")).read())
    for n in ast.walk(tree):
        if isinstance(n, ast.ImportFrom) and n.module == "train":
            return [a.name for a in n.names]
    return []


if __name__ == "__main__":
    data = {"flags": flags_of(os.path.join(REF, "train.py")), "launch_lines": launch_lines(), "infer_imports": infer_imports()}
    json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)
    print(OUT, len(data["flags"]), "flags,", len(data["launch_lines"]), "launch lines,", data["infer_imports"])
