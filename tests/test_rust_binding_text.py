"""The Rust declarations a maintainer of the reference would bind (rust/raytracer-hip-sys/src/lib.rs) cannot be
compiled here (no cargo / rustc in the image), so their text is held to the C header instead: every struct of
include/rt/abi.h must appear with the same fields in the same order and of the matching type, every function with
the same parameters and return type, and nothing may be declared on the Rust side that the header does not have."""
import os
import re

from conftest import ROOT

C2RUST = {"uint8_t": "u8", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "float": "f32", "double": "f64", "int": "c_int",
          "char": "c_char", "void": "c_void"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_structs(text):
    out = {}
    for body, name in re.findall(r"typedef struct \w+ \{(.*?)\} (\w+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(const )?(\w+)\s*(\*?)\s*(.*)", decl)
            const, ctype, star, names = m.groups()
            for nm in names.split(","):
                nm = nm.strip()
                dims = [int(d) for d in re.findall(r"\[(\d+)\]", nm)]
                nm = nm.split("[")[0].strip()
                ptr = star
                if nm.startswith("*"):
                    ptr, nm = "*", nm[1:].strip()
                rt = C2RUST.get(ctype, ctype)
                for d in reversed(dims):
                    rt = "[%s; %d]" % (rt, d)
                if ptr:
                    rt = ("*const " if const else "*mut ") + rt
                fields.append((nm, rt))
        out[name] = fields
    return out


def c_type_to_rust(t):
    t = " ".join(t.replace("*", " * ").split())
    const = t.startswith("const ")
    if const:
        t = t[6:]
    parts = t.split()
    base = C2RUST.get(parts[0], parts[0])
    stars = parts.count("*")
    if stars == 0:
        return base
    r = ("*const " if const else "*mut ") + base
    for _ in range(stars - 1):
        r = "*mut " + r
    return r


def c_functions(text):
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const )?\w+\s*\*?)\s*(rt_\w+)\s*\(([^)]*)\)\s*;", text, flags=re.M):
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                m = re.match(r"(.*?)(\w+)$", a)
                params.append((m.group(2), c_type_to_rust(m.group(1))))
        out[name] = (params, None if ret.strip() == "void" else c_type_to_rust(ret))
    return out


def rust_structs(text):
    out = {}
    for name, body in re.findall(r"pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        depth, cur = 0, ""
        for ch in body + ",":
            if ch in "[<(":
                depth += 1
            if ch in "]>)":
                depth -= 1
            if ch == "," and depth == 0:
                cur = cur.strip()
                if cur:
                    nm, ty = cur.replace("pub ", "", 1).split(":", 1)
                    fields.append((nm.strip(), " ".join(ty.split())))
                cur = ""
            else:
                cur += ch
        out[name] = fields
    return out


def rust_functions(text):
    out = {}
    for name, args, ret in re.findall(r"pub fn (rt_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text):
        params = []
        for a in args.split(","):
            a = a.strip()
            if a:
                nm, ty = a.split(":", 1)
                params.append((nm.strip(), " ".join(ty.split())))
        out[name] = (params, ret.strip() if ret else None)
    return out


def test_rust_declarations_match_the_header():
    header = strip_comments(open(os.path.join(ROOT, "include", "rt", "abi.h")).read())
    rust = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "raytracer-hip-sys", "src", "lib.rs")).read())
    cs, rs = c_structs(header), rust_structs(rust)
    assert set(cs) == set(rs), (sorted(set(cs) - set(rs)), sorted(set(rs) - set(cs)))
    for name in cs:
        assert cs[name] == rs[name], (name, cs[name], rs[name])
    cf, rf = c_functions(header), rust_functions(rust)
    assert set(cf) == set(rf), (sorted(set(cf) - set(rf)), sorted(set(rf) - set(cf)))
    for name in cf:
        assert cf[name] == rf[name], (name, cf[name], rf[name])
    assert len(cf) >= 24 and len(cs) >= 21
    # constants both sides name
    for cname, pattern in (("RT_BVH_TOKEN_LEAF", r"RT_BVH_TOKEN_LEAF 0xFu"), ("RT_BVH_NOALLOC_MAX_OBJECTS", r"RT_BVH_NOALLOC_MAX_OBJECTS 65536u"),
                           ("RT_DESC_BOXES_MAY_NOT_BOUND", r"RT_DESC_BOXES_MAY_NOT_BOUND 1u"),
                           ("RT_INNER_CARRIED", r"RT_INNER_CARRIED 1u"), ("RT_INNER_COVER", r"RT_INNER_COVER 2u")):
        assert re.search(pattern, header) and cname in rust, cname
        value = int(re.search(r"#define %s (0x[0-9A-Fa-f]+|\d+)u" % cname, header).group(1), 0)
        assert int(re.search(r"pub const %s: \w+ = (0x[0-9A-Fa-f]+|\d+);" % cname, rust).group(1), 0) == value, cname
    # the opaque handles
    for h in ("RtScene", "RtShardedScene"):
        assert re.search(r"typedef struct %s %s;" % (h, h), header) and re.search(r"pub enum %s \{\}" % h, rust)


# ---- rust/reference-patch/flatten.rs: the patch a maintainer would add to the reference's crate -------------------------
FLATTEN = os.path.join(ROOT, "rust", "reference-patch", "flatten.rs")
REFERENCE_SRC = "/root/reference/src"


def rust_code(path):
    return re.sub(r"//[^\n]*", "", open(path).read())


def struct_literals(text):
    """(struct name, [field names]) of every `RtName { ... }` literal (not definitions, not patterns with `..`)."""
    out = []
    for m in re.finditer(r"\b(Rt[A-Z]\w*)\s*\{", text):
        before = text[max(0, m.start() - 12):m.start()]
        if re.search(r"(struct|impl|for|->)\s+$", before):  # a definition or a return type, not a literal
            continue
        depth, i, cur, fields = 1, m.end(), "", []
        while depth:
            ch = text[i]
            if ch in "{[(":
                depth += 1
            elif ch in "}])":
                depth -= 1
            if (ch == "," and depth == 1) or depth == 0:
                cur = cur.strip()
                if cur:
                    fields.append(re.match(r"(\w+)", cur).group(1))
                cur = ""
            else:
                cur += ch
            i += 1
        out.append((m.group(1), fields))
    return out


def header_constants(header):
    c = {name: int(v, 0) for name, v in re.findall(r"#define (RT_\w+) (0x[0-9A-Fa-f]+|\d+)u", header)}
    for body in re.findall(r"enum \w+ \{(.*?)\};", header, flags=re.S):
        for name, v in re.findall(r"(RT_\w+) = (-?\d+)", body):
            c[name] = int(v)
    return c


def test_flatten_patch_builds_the_header_structs():
    header = strip_comments(open(os.path.join(ROOT, "include", "rt", "abi.h")).read())
    rust = rust_code(FLATTEN)
    cs = c_structs(header)
    lits = struct_literals(rust)
    assert len(lits) >= 20
    seen = set()
    for name, fields in lits:
        assert name in cs, name
        assert fields == [f for f, _ in cs[name]], (name, fields, [f for f, _ in cs[name]])  # every field, in the header's order
        seen.add(name)
    # every table of the scene description is filled from a literal of its row type
    for row in ("RtNode", "RtSphere", "RtMovingSphere", "RtRect", "RtBox6", "RtTriangle", "RtXform", "RtMedium", "RtMaterial", "RtTexture",
                "RtPerlin", "RtImage", "RtSceneDesc", "RtCamera", "RtRenderParams"):
        assert row in seen, row
    # the constants the patch spells out are the header's
    consts = header_constants(header)
    mine = re.findall(r"^const (RT_\w+): u32 = ([0-9A-Fa-fx_]+);", rust, flags=re.M)
    assert len(mine) >= 12
    for name, value in mine:
        assert name in consts and int(value.replace("_", ""), 0) == consts[name], (name, value, consts.get(name))
    # what it passes to the library is declared by the binding crate
    sys_rust = rust_code(os.path.join(ROOT, "rust", "raytracer-hip-sys", "src", "lib.rs"))
    declared = set(rust_functions(sys_rust))
    for fn in set(re.findall(r"\b(rt_\w+)\s*\(", rust)):
        assert fn in declared, fn


def test_flatten_patch_names_what_the_reference_has():
    """Every item the patch imports from the reference's crate exists there under that name and kind, the private fields
    it reads are fields of those structs, and every `src/file.rs:line` it cites lies inside that file. Needs the
    reference tree (present where the package is built, absent on the GPU box)."""
    if not os.path.isdir(REFERENCE_SRC):
        import pytest
        pytest.skip("the reference tree is not on this machine")
    text = open(FLATTEN).read()
    rust = rust_code(FLATTEN)
    for module, items in re.findall(r"use crate::(\w+)::\{?([\w, ]+)\}?;", rust):
        path = os.path.join(REFERENCE_SRC, module + ".rs")
        assert os.path.exists(path), module
        src = open(path).read()
        for item in (i.strip() for i in items.split(",")):
            assert re.search(r"pub (struct|trait|enum|fn) %s\b" % item, src), (module, item)
    for name, line in re.findall(r"src/(\w+\.rs):(\d+)", text):
        path = os.path.join(REFERENCE_SRC, name)
        assert os.path.exists(path), name
        assert 1 <= int(line) <= len(open(path).read().splitlines()), (name, line)
    # fields read through a typed binding `x: &Type` (x.field) must be fields of the reference's struct
    checked = 0
    functions = []
    for m in re.finditer(r"pub fn \w+\(([^)]*)\)[^{;]*\{", rust):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(rust[i], 0)
            i += 1
        functions.append((m.group(1), rust[m.end():i - 1]))
    for fn_args, body in functions:
        for var, ty in re.findall(r"(\w+): &(\w+)", fn_args):
            hits = [m for m in os.listdir(REFERENCE_SRC) if re.search(r"pub struct %s\b" % ty, open(os.path.join(REFERENCE_SRC, m)).read())]
            if not hits:
                continue
            src = open(os.path.join(REFERENCE_SRC, hits[0])).read()
            decl = re.search(r"pub struct %s\s*\{(.*?)\}" % ty, src, flags=re.S)
            if not decl:
                continue
            fields = set(re.findall(r"(?:pub )?(\w+)\s*:", decl.group(1)))
            for used in set(re.findall(r"\b%s\.(\w+)\b(?!\s*\()" % var, body)):
                assert used in fields, (ty, used, sorted(fields))
                checked += 1
    assert checked >= 20
