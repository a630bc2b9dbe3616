"""Extracts the constants of SURVEY.md Appendix B from the reference's own sources into tests/golden/constants_ref.json (VALUES only, no source text).

Run in the build container (the reference is mounted at /root/reference there; it does not travel to the GPU box, the JSON does):

    python tests/golden/make_constants_ref.py            # rewrites tests/golden/constants_ref.json
    python tests/golden/make_constants_ref.py --check    # exit 1 if the committed file differs from a fresh extraction

Sources (relative to /root/reference/src):
  util/settings.cpp        every `int|float|double|bool name = expr;` definition (setting_*, sparsityFactor, ...), staticPattern[8] (the pattern in use)
  util/settings.h          #define PYR_LEVELS / patternNum / patternPadding / SOLVER_*
  util/NumType.h           #define MAX_RES_PER_POINT / NUM_THREADS / CPARS
  FullSystem/HessianBlocks.h   #define SCALE_* ; the constructor's `frameEnergyTH = 8*8*patternNum`

Every value is evaluated with C semantics for the declared type: an expression is computed in double unless an operand carries the f suffix, then stored
into the declared type (`float setting_initialRotPrior = 1e11;` is 99999997952.0, the value the reference's solver sees)."""
import ast
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("NALO_REFERENCE", "/root/reference") + "/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants_ref.json")


def c_eval(expr, macros):
    """a C constant expression of ints / floats / doubles: literals, + - * /, |, parentheses, (int) casts, known macros"""
    e = expr.strip()
    e = re.sub(r"\(int\)", "", e)
    e = re.sub(r"\btrue\b", "1", e)
    e = re.sub(r"\bfalse\b", "0", e)
    for k in sorted(macros, key=len, reverse=True):
        e = re.sub(r"\b%s\b" % re.escape(k), "(%s)" % repr(macros[k]), e)
    # tag literals: 1.5f -> F(1.5), 1.5 / 1e11 -> D(..), integers stay Python ints
    def lit(m):
        t = m.group(0)
        if t[-1] in "fF":
            return "F(%s)" % t[:-1]
        if "." in t or "e" in t.lower():
            return "D(%s)" % t
        return t
    e = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+|\d+)[fF]?(?![\w.])", lit, e)
    tree = ast.parse(e, mode="eval")
    for node in ast.walk(tree):
        if not isinstance(node, (ast.Expression, ast.BinOp, ast.UnaryOp, ast.Constant, ast.Call, ast.Name, ast.Load, ast.Add, ast.Sub, ast.Mult, ast.Div, ast.BitOr,
                                 ast.USub, ast.UAdd)):
            raise ValueError("unsupported expression: %r" % expr)
        if isinstance(node, ast.Name) and node.id not in ("F", "D"):
            raise ValueError("unknown identifier %r in %r" % (node.id, expr))

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant):
            return n.value
        if isinstance(n, ast.Call):
            return (np.float32 if n.func.id == "F" else np.float64)(ev(n.args[0]))
        if isinstance(n, ast.UnaryOp):
            v = ev(n.operand)
            return -v if isinstance(n.op, ast.USub) else v
        a, b = ev(n.left), ev(n.right)
        if isinstance(n.op, ast.BitOr):
            return int(a) | int(b)
        # the usual arithmetic conversions: double beats float beats int
        if isinstance(a, np.float64) or isinstance(b, np.float64):
            a, b = np.float64(a), np.float64(b)
        elif isinstance(a, np.float32) or isinstance(b, np.float32):
            a, b = np.float32(a), np.float32(b)
        if isinstance(n.op, ast.Add):
            return a + b
        if isinstance(n.op, ast.Sub):
            return a - b
        if isinstance(n.op, ast.Mult):
            return a * b
        if isinstance(a, int) and isinstance(b, int):
            return int(a / b)                      # C integer division truncates
        return a / b
    return ev(tree)


def store(ctype, v):
    if ctype == "float":
        return float(np.float32(v))
    if ctype == "double":
        return float(np.float64(v))
    return int(v)                                   # int, bool


def lines(path):
    with open(os.path.join(REF, path), "r", errors="replace") as f:
        return f.read().split("\n")


def extract():
    out = {}

    def put(name, ctype, value, src):
        out[name] = {"type": ctype, "value": value, "src": src}

    macros = {}
    for path, want in (("util/settings.h", ("PYR_LEVELS", "patternNum", "patternPadding", "SOLVER_")), ("util/NumType.h", ("MAX_RES_PER_POINT", "NUM_THREADS", "CPARS")),
                       ("FullSystem/HessianBlocks.h", ("SCALE_",))):
        for i, ln in enumerate(lines(path)):
            m = re.match(r"\s*#define\s+(\w+)\s+(.+?)\s*(//.*)?$", ln)
            if not m or not any(m.group(1) == w or (w.endswith("_") and m.group(1).startswith(w)) for w in want):
                continue
            name, expr = m.group(1), m.group(2)
            if name.endswith("_INVERSE"):
                continue
            v = c_eval(expr, macros)
            ctype = "float" if isinstance(v, np.float32) else ("double" if isinstance(v, np.float64) else "int")
            macros[name] = store(ctype, v) if ctype == "int" else float(v)
            put(name, ctype, store(ctype, v), "%s:%d" % (path, i + 1))
    # settings.cpp: scalar definitions
    src = lines("util/settings.cpp")
    for i, ln in enumerate(src):
        m = re.match(r"\s*(int|float|double|bool)\s+(\w+)\s*=\s*([^;{]+);", ln)
        if not m:
            continue
        ctype, name, expr = m.groups()
        try:
            v = c_eval(expr, macros)
        except (ValueError, SyntaxError):
            continue                                # not a constant expression of this kind
        put(name, ctype, store(ctype, v), "util/settings.cpp:%d" % (i + 1))
    # staticPattern[patternIdx][k] for the pattern `#define patternP staticPattern[8]` selects
    pat_idx = None
    for ln in lines("util/settings.h"):
        m = re.match(r"\s*#define\s+patternP\s+staticPattern\[(\d+)\]", ln)
        if m:
            pat_idx = int(m.group(1))
    text = "\n".join(src)
    start = text.index("int staticPattern[")
    body = text[text.index("{", start):]
    body = re.sub(r"//[^\n]*", "", body)
    depth, groups, cur = 0, [], None
    for ch in body:
        if ch == "{":
            depth += 1
            if depth == 2:
                cur = ""
                continue
        elif ch == "}":
            depth -= 1
            if depth == 1:
                groups.append(cur)
                cur = None
                continue
            if depth == 0:
                break
        if cur is not None:
            cur += ch
    pairs = re.findall(r"\{\s*([-+]?\d+)\s*,\s*([-+]?\d+)\s*\}", groups[pat_idx])
    line_of = next(i + 1 for i, ln in enumerate(src) if "int staticPattern[" in ln)
    for k in range(out["patternNum"]["value"]):
        put("patternP[%d].x" % k, "int", int(pairs[k][0]), "util/settings.cpp:%d (staticPattern[%d])" % (line_of, pat_idx))
        put("patternP[%d].y" % k, "int", int(pairs[k][1]), "util/settings.cpp:%d (staticPattern[%d])" % (line_of, pat_idx))
    # FrameHessian(): frameEnergyTH = 8*8*patternNum
    for i, ln in enumerate(lines("FullSystem/HessianBlocks.h")):
        m = re.match(r"\s*frameEnergyTH\s*=\s*([^;]+);", ln)
        if m:
            put("frameEnergyTH_init", "float", store("float", c_eval(m.group(1), macros)), "FullSystem/HessianBlocks.h:%d" % (i + 1))
            break
    return out


def main():
    data = {"comment": "constants of the reference, extracted by tests/golden/make_constants_ref.py from /root/reference/src (values only)", "constants": extract()}
    text = json.dumps(data, indent=1, sort_keys=True) + "\n"
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text
        print("constants_ref.json is %s" % ("up to date" if ok else "STALE"))
        sys.exit(0 if ok else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote %s: %d constants" % (OUT, len(data["constants"])))


if __name__ == "__main__":
    main()
