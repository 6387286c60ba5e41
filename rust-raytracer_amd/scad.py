"""Reader for the OpenSCAD dialect the reference's front end accepts (crates/openscad): turns `.scad` text
into calls on a `bindings.Scene`, for either backend, the way `run_openscad` (crates/openscad/src/lib.rs:88-128)
turns it into a `SceneData`.

It follows the reference's tokenizer (tokenizer.rs:230-531), parser (parser.rs:355-1134, operator
precedences parser.rs:178-193) and interpreter (interpreter/mod.rs:162-585, expr.rs, functions.rs,
modules.rs), including their quirks:

* unary minus takes a whole expression as its operand (parser.rs:841-856): `-a + b` is `-(a + b)`;
* there is one variable scope for the whole file (blocks do not open one; only user function calls do,
  mod.rs:544-570, functions.rs:598-606);
* `for` ranges exclude their end (mod.rs:289-302);
* vectors that become positions are axis-swapped, OpenSCAD [x, y, z] -> (-x, z, y) (value.rs:80-105), and a
  bare number becomes (-v, v, v) (value.rs:52); colours are not swapped (value.rs:107-131);
* vector * vector is element-wise, truncated to the shorter operand (expr.rs:490-511);
* `rotate([x, y, z])` chains rotate_x / rotate_y / rotate_z of the swapped components, skipping zeros
  (modules.rs:288-326);
* `rands` draws from MT19937-64 with its default seed 5489 (`Mt64::new_unseeded`, mod.rs:157) and scales by
  u64::MAX (functions.rs:534-581);
* the world is one BoundingVolumeHierarchy over the top-level nodes and there are never lights
  (mod.rs:188-196).

Perlin tables come from the scene generator of `scenes.py` (the reference draws them from an unseeded
thread RNG, functions.rs:506-510); image files are resolved by a caller-supplied loader.
Constructs the reference's interpreter answers with `todo!()` raise ScadError here.
"""
import math

import numpy as np

from . import scenes as _scenes


class ScadError(Exception):
    pass


# ---------------------------------------------------------------------------------------------- tokenizer
_SINGLE = {"(": "(", ")": ")", "[": "[", "]": "]", "{": "{", "}": "}", ",": ",", ":": ":", ";": ";", ".": ".", "+": "+", "-": "-",
           "*": "*", "?": "?", "%": "%", "^": "^"}
_KEYWORDS = {"for", "if", "else", "true", "false"}


def tokenize(src: str):
    """-> (tokens, messages); a token is (kind, value).  tokenizer.rs:289-531."""
    toks, msgs = [], []
    i, n = 0, len(src)
    while True:
        while i < n and src[i].isspace():
            i += 1
        if i >= n:
            break
        ch = src[i]
        if ch == "/":
            if i + 1 < n and src[i + 1] == "/":
                j = src.find("\n", i)
                i = n if j < 0 else j
                continue
            if i + 1 < n and src[i + 1] == "*":
                j = src.find("*/", i + 2)
                i = n if j < 0 else j + 2
                continue
            toks.append(("/", None))
            i += 1
        elif ch in _SINGLE:
            toks.append((ch, None))
            i += 1
        elif ch in "&|":
            if i + 1 < n and src[i + 1] == ch:
                toks.append((ch + ch, None))
                i += 2
            else:
                msgs.append(("error", f"Invalid character '{src[i + 1] if i + 1 < n else ch}'"))
                i += 1
        elif ch in "!=<>":
            if i + 1 < n and src[i + 1] == "=":
                toks.append((ch + "=", None))
                i += 2
            else:
                toks.append((ch, None))
                i += 1
        elif ch == '"':
            j = i + 1
            out = []
            while j < n and src[j] != '"':
                if src[j] == "\\" and j + 1 < n:
                    out.append({"n": "\n", "t": "\t", "r": "\r", '"': '"', "\\": "\\"}.get(src[j + 1], src[j + 1]))
                    j += 2
                else:
                    out.append(src[j])
                    j += 1
            toks.append(("string", "".join(out)))
            i = j + 1
        elif ch.isalpha() or ch in "_$":
            j = i
            while j < n and (src[j].isalnum() or src[j] in "_$"):
                j += 1
            ident = src[i:j]
            i = j
            if ident in ("include", "use"):
                while i < n and src[i].isspace():
                    i += 1
                if i >= n or src[i] != "<":
                    raise ScadError("include should be followed by '<'")
                j = src.find(">", i)
                if j < 0:
                    raise ScadError("unterminated include")
                toks.append((ident, src[i + 1:j]))
                i = j + 1
            elif ident in _KEYWORDS:
                toks.append((ident, None))
            else:
                toks.append(("ident", ident))
        else:
            # number: digits with at most one '.', then an optional exponent (tokenizer.rs:230-287)
            j, seen_digit, seen_dot = i, False, False
            while j < n and (src[j].isdigit() and src[j].isascii() or (src[j] == "." and not seen_dot)):
                if src[j] == ".":
                    seen_dot = True
                else:
                    seen_digit = True
                j += 1
            if not seen_digit:
                msgs.append(("error", f"Invalid character '{ch}'"))
                i += 1
                continue
            if j < n and src[j] in "eE":
                k = j + 1
                if k < n and src[k] in "+-":
                    k += 1
                while k < n and src[k].isdigit():
                    k += 1
                j = k
            try:
                toks.append(("number", float(src[i:j])))
            except ValueError:
                msgs.append(("error", f"Invalid character '{ch}'"))
                i += 1
                continue
            i = j
    toks.append(("eof", None))
    return toks, msgs


# ---------------------------------------------------------------------------------------------- parser
_PREC = {"||": 0, "&&": 1, "<": 2, "<=": 2, ">": 2, ">=": 2, "==": 2, "!=": 2, "+": 3, "-": 3, "*": 4, "/": 4, "%": 4, "^": 5}


class _Parser:
    def __init__(self, toks):
        self.t = toks
        self.p = 0
        self.messages = []

    def cur(self):
        return self.t[self.p]

    def peek(self, k=1):
        return self.t[min(self.p + k, len(self.t) - 1)]

    def adv(self):
        tok = self.t[self.p]
        if self.p < len(self.t) - 1:
            self.p += 1
        return tok

    def expect(self, kind):
        if self.cur()[0] != kind:
            found = self.adv()
            raise ScadError(f"Expected {kind} but found {found[0]}")
        return self.adv()

    def parse(self):
        stmts = []
        while self.cur()[0] != "eof":
            at = self.p
            try:
                stmts.append(self.statement())
            except ScadError as e:  # parser.rs:949-952: the message is recorded and parsing goes on
                self.messages.append(("error", str(e)))
                if self.p == at:
                    self.adv()
        return stmts

    def statement(self):
        k, v = self.cur()
        if k == ";":
            self.adv()
            return ("empty",)
        if k == "include":
            self.adv()
            return ("include", v)
        if k == "use":
            raise ScadError("use is not supported")
        if k == "ident" and v == "function":
            self.adv()
            return self.function_decl()
        if k == "ident" and v == "module":
            raise ScadError("module declarations are not supported (parser.rs:394-397)")
        if k == "ident" and self.peek()[0] == "=":
            name = self.adv()[1]
            self.expect("=")
            e = self.expr()
            self.expect(";")
            return ("assign", name, e)
        return self.module_instantiation()

    def module_instantiation(self):
        if self.cur()[0] == "if":
            self.adv()
            self.expect("(")
            cond = self.expr()
            self.expect(")")
            t = self.child_statements()
            f = []
            if self.cur()[0] == "else":
                self.adv()
                f = self.child_statements()
            return ("if", cond, t, f)
        k, v = self.cur()
        if k == "for":
            name = "for"
        elif k == "ident":
            name = v
        else:
            self.adv()
            raise ScadError(f"Expected for or identifier but found: {k}")
        self.adv()
        args = self.call_arguments()
        kids = self.child_statements()
        return ("module", name, args, kids)

    def child_statements(self):
        if self.cur()[0] == ";":
            self.adv()
            return []
        if self.cur()[0] == "{":
            self.adv()
            out = []
            while self.cur()[0] != "}":
                if self.cur()[0] == "eof":
                    raise ScadError("missing }")
                out.append(self.statement())
            self.adv()
            return out
        return [self.module_instantiation()]

    def call_arguments(self):
        self.expect("(")
        args = []
        while self.cur()[0] != ")":
            if self.cur()[0] == "eof":
                raise ScadError("missing )")
            if self.cur()[0] == "ident" and self.peek()[0] == "=":
                name = self.adv()[1]
                self.adv()
                args.append((name, self.expr()))
            else:
                args.append((None, self.expr()))
            while self.cur()[0] == ",":
                self.adv()
        self.adv()
        return args

    def function_decl(self):
        name = self.expect("ident")[1]
        self.expect("(")
        params = []
        while self.cur()[0] == "ident":
            params.append(self.adv()[1])
            if self.cur()[0] == "=":
                raise ScadError("declared arguments with defaults are not supported (parser.rs:1083-1085)")
            if self.cur()[0] == ")":
                break
            while self.cur()[0] == ",":
                self.adv()
        self.expect(")")
        self.expect("=")
        e = self.expr()
        self.expect(";")
        return ("function", name, params, e)

    def expr(self):
        lhs = self.binary(0)
        if self.cur()[0] == "?":
            self.adv()
            a = self.binary(0)
            self.expect(":")
            b = self.binary(0)
            return ("ternary", lhs, a, b)
        return lhs

    def binary(self, min_prec):
        lhs = self.unary()
        while self.cur()[0] in _PREC and _PREC[self.cur()[0]] >= min_prec:
            op = self.adv()[0]
            rhs = self.binary(_PREC[op] + 1)
            lhs = ("binary", op, lhs, rhs)
        return lhs

    def unary(self):
        k, v = self.cur()
        if k == "[":
            self.adv()
            items, comma, colon = [], False, False
            while self.cur()[0] != "]":
                if self.cur()[0] == "eof":
                    raise ScadError("missing ]")
                if not colon and self.cur()[0] == ",":
                    comma = True
                    self.adv()
                    continue
                if not comma and self.cur()[0] == ":":
                    if not items:
                        raise ScadError("leading colon is not allowed")
                    colon = True
                    self.adv()
                    continue
                items.append(self.expr())
            self.adv()
            if colon:
                if len(items) == 2:
                    lhs = ("range", items[0], items[1], None)
                elif len(items) == 3:
                    lhs = ("range", items[0], items[2], items[1])
                else:
                    raise ScadError("a range takes 2 or 3 expressions")
            else:
                lhs = ("vector", items)
        elif k == "true":
            self.adv()
            lhs = ("bool", True)
        elif k == "false":
            self.adv()
            lhs = ("bool", False)
        elif k == "ident":
            self.adv()
            if self.cur()[0] == "(":
                lhs = ("call", v, self.call_arguments())
            else:
                lhs = ("name", v)
        elif k == "number":
            self.adv()
            lhs = ("number", v)
        elif k == "string":
            self.adv()
            lhs = ("string", v)
        elif k == "-":
            self.adv()
            lhs = ("neg", self.expr())
        elif k == "!":
            self.adv()
            lhs = ("not", self.expr())
        elif k == "(":
            self.adv()
            lhs = self.expr()
            self.expect(")")
        else:
            self.adv()
            raise ScadError(f"unhandled: {k}")
        while True:
            if self.cur()[0] == "[":
                self.adv()
                idx = self.expr()
                self.expect("]")
                lhs = ("index", lhs, idx)
            elif self.cur()[0] == ".":
                self.adv()
                lhs = ("field", lhs, self.expect("ident")[1])
            else:
                return lhs


def parse(src: str):
    toks, msgs = tokenize(src)
    p = _Parser(toks)
    stmts = p.parse()
    return stmts, msgs + p.messages


# ---------------------------------------------------------------------------------------------- values
class Undef:
    def __repr__(self):
        return "undef"


UNDEF = Undef()


class Range:
    def __init__(self, start, end, increment):
        self.start, self.end, self.increment = start, end, increment


class TextureRef:
    def __init__(self, tex_id):
        self.id = tex_id


class FunctionRef:
    def __init__(self, name):
        self.name = name


def _is_num(v):
    return isinstance(v, float)


def format_value(v) -> str:
    """Display for Value, value.rs:153-185."""
    if isinstance(v, bool):
        return "true" if v else "false"
    if _is_num(v):
        if math.isnan(v):
            return "NaN"
        if math.isinf(v):
            return "inf" if v > 0 else "-inf"
        return ("%.6f" % v).rstrip("0").rstrip(".")
    if isinstance(v, str):
        out = ['"']
        for ch in v:
            out.append({"\n": "\\n", "\t": "\\t", "\r": "\\r", '"': '\\"', "\\": "\\\\"}.get(ch, ch))
        out.append('"')
        return "".join(out)
    if isinstance(v, list):
        return "[" + ", ".join(format_value(x) for x in v) + "]"
    if v is UNDEF:
        return "undef"
    if isinstance(v, FunctionRef):
        return f"fn({v.name})"
    raise ScadError(f"cannot print {type(v).__name__}")


def _truthy(v):
    if isinstance(v, bool):
        return v
    if _is_num(v):
        return v != 0.0
    if v is UNDEF:
        return False
    if isinstance(v, FunctionRef):
        return True
    raise ScadError(f"is_truthy of {type(v).__name__}")


def _num(v):
    if not _is_num(v):
        raise ScadError(f"to_number {format_value(v) if not isinstance(v, (Range, TextureRef)) else type(v).__name__}")
    return v


def _rust_u32(x: float) -> int:  # `as u32`: saturating, NaN -> 0
    if math.isnan(x):
        return 0
    return int(min(max(x, 0.0), 4294967295.0))


def _rust_i64(x: float) -> int:
    if math.isnan(x):
        return 0
    return int(min(max(x, -9.223372036854775808e18), 9.223372036854775807e18))


def to_vector3(v):
    """value.rs:50-56, 80-105: OpenSCAD [x, y, z] -> (-x, z, y)."""
    if _is_num(v):
        return (-v, v, v)
    if isinstance(v, list) and len(v) == 3 and all(_is_num(x) for x in v):
        return (-v[0], v[2], v[1])
    raise ScadError("expected a vector of 3 numbers")


def to_color(v):
    if _is_num(v):
        return (v, v, v)
    if isinstance(v, list) and len(v) == 3 and all(_is_num(x) for x in v):
        return (v[0], v[1], v[2])
    raise ScadError("expected a colour of 3 numbers")


class Mt64:
    """MT19937-64 with the default seed 5489 (rand_mt 5.0.0 `Mt64::new_unseeded`)."""
    NN, MM = 312, 156

    def __init__(self, seed=5489):
        m = (1 << 64) - 1
        mt = [seed & m]
        for i in range(1, self.NN):
            mt.append((6364136223846793005 * (mt[-1] ^ (mt[-1] >> 62)) + i) & m)
        self.mt, self.i = mt, self.NN

    def next_u64(self) -> int:
        if self.i >= self.NN:
            mt, NN, MM = self.mt, self.NN, self.MM
            for k in range(NN):
                x = (mt[k] & 0xFFFFFFFF80000000) | (mt[(k + 1) % NN] & 0x7FFFFFFF)
                mt[k] = mt[(k + MM) % NN] ^ (x >> 1) ^ (0xB5026F5AA96619E9 if x & 1 else 0)
            self.i = 0
        x = self.mt[self.i]
        self.i += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x


# ---------------------------------------------------------------------------------------------- interpreter
_MATERIAL_MODULES = ("color", "lambertian", "dielectric", "metal", "diffuse_light")


class Interpreter:
    def __init__(self, scene, scene_seed=1, image_loader=None):
        self.sc = scene
        self.vars = [{"$fn": 0.0, "$fs": 2.0, "$fa": 12.0, "$t": 0.0, "$vpr": [55.0, 0.0, 25.0], "$vpt": [0.0, 0.0, 0.0], "$vpd": 140.0,
                      "$vpf": 22.5, "$preview": True}]
        self.functions = {}
        self.material_stack = []
        self.messages = []
        self.world = []
        self.camera = None
        self.rng = Mt64()
        self.scene_rng = _scenes.SceneRandom(scene_seed)
        self.image_loader = image_loader
        self._mat_cache = {}

    # -- statements (mod.rs:204-232)
    def run(self, statements):
        for st in statements:
            try:
                self.world += self.statement(st)
            except ScadError as e:
                self.messages.append(("error", str(e)))
        cam = self.camera or dict(aspect_ratio=1.0, image_width=600, samples_per_pixel=10, max_depth=50, defocus_angle=0.0,
                                  background=(0.7, 0.8, 1.0), look_at=(0.0, 0.0, 0.0), look_from=(-50.0, 70.0, -50.0), up=(0.0, 1.0, 0.0))
        self.sc.set_world(self.sc.bvh(self.world))
        self.sc.set_camera(**cam)
        self.sc.commit()
        return self.sc, self.messages

    def statement(self, st):
        k = st[0]
        if k == "empty":
            return []
        if k == "module":
            return self.module(st[1], st[2], st[3])
        if k == "assign":
            v = self.eval(st[2])
            if st[1].startswith("$") and not _is_num(v):
                raise ScadError(f"expected number for {st[1]}")
            self.vars[-1][st[1]] = v
            return []
        if k == "include":
            if st[1].endswith("caustic.scad"):
                return []
            raise ScadError(f"include {st[1]}")
        if k == "function":
            self.functions[st[1]] = (st[2], st[3])
            return []
        if k == "if":
            return self.children(st[2] if _truthy(self.eval(st[1])) else st[3])
        raise ScadError(f"statement {k}")

    def children(self, stmts):
        out = []
        for s in stmts:
            out += self.statement(s)
        return out

    # -- arguments (mod.rs:345-389)
    def convert_args(self, names, args):
        out, named = {}, False
        for pos, (name, e) in enumerate(args):
            if name is None:
                if named:
                    raise ScadError("no positional arguments after a named one")
                if pos >= len(names):
                    raise ScadError(f"argument {pos} past the end of {names}")
                out[names[pos]] = self.eval(e)
            else:
                named = True
                if name not in names:
                    raise ScadError(f"unknown arg name: {name}")
                out[name] = self.eval(e)
        return out

    def current_material(self):
        if self.material_stack:
            return self.material_stack[-1]
        return self._material(("lambertian", 0.99, 0.85, 0.26))  # mod.rs:234-240

    def _material(self, key):
        m = self._mat_cache.get(key)
        if m is None:
            kind = key[0]
            if kind == "lambertian":
                m = self.sc.lambertian_color(*key[1:])
            elif kind == "metal":
                m = self.sc.metal(*key[1:])
            elif kind == "dielectric":
                m = self.sc.dielectric(key[1])
            elif kind == "light":
                m = self.sc.diffuse_light_color(*key[1:])
            else:
                m = self.sc.lambertian_tex(key[1])
            self._mat_cache[key] = m
        return m

    # -- modules (modules.rs:17-74)
    def module(self, name, args, kids):
        if name in _MATERIAL_MODULES:
            self.material_stack.append(self.make_material(name, args))
        elif name == "for":
            return self.for_loop(args, kids)
        nodes = self.children(kids)
        if name in _MATERIAL_MODULES:
            self.material_stack.pop()
            return nodes
        if name in ("circle", "cube", "sphere", "cylinder", "quad", "camera", "echo") and nodes:
            raise ScadError(f"{name} should not have children")
        if name in ("translate", "rotate", "scale") and not nodes:
            raise ScadError(f"{name} should have children")
        sc = self.sc
        if name == "circle":  # modules.rs:76-105
            a = self.convert_args(["r", "d"], args)
            r = 1.0
            if "r" in a:
                r = _num(a["r"])
            if "d" in a:
                r = _num(a["d"]) / 2.0
            return [sc.disc((0.0, 0.0, 0.0), r, (0.0, 1.0, 0.0), self.current_material())]
        if name == "cube":  # modules.rs:107-137
            a = self.convert_args(["size", "center"], args)
            size = to_vector3(a["size"]) if "size" in a else (0.0, 0.0, 0.0)
            center = False
            if "center" in a:
                if not isinstance(a["center"], bool):
                    raise ScadError("center must be a boolean")
                center = a["center"]
            lo, hi = (0.0, 0.0, 0.0), size
            if center:
                lo = tuple(l - s / 2.0 for l, s in zip(lo, size))
                hi = tuple(h - s / 2.0 for h, s in zip(hi, size))
            return [sc.box(lo, hi, self.current_material())]
        if name == "sphere":  # modules.rs:139-163
            a = self.convert_args(["r", "d"], args)
            r = 1.0
            if "r" in a:
                r = _num(a["r"])
            elif "d" in a:
                r = _num(a["d"]) / 2.0
            return [sc.sphere((0.0, 0.0, 0.0), r, self.current_material())]
        if name == "cylinder":  # modules.rs:165-232
            a = self.convert_args(["h", "r1", "r2", "center", "r", "d", "d1", "d2"], args)
            h = _num(a["h"]) if "h" in a else 1.0
            r1 = _num(a["r1"]) if "r1" in a else 1.0
            r2 = _num(a["r2"]) if "r2" in a else 1.0
            if "r" in a:
                r1 = r2 = _num(a["r"])
            if "d1" in a:
                r1 = _num(a["d1"]) / 2.0
            if "d2" in a:
                r2 = _num(a["d2"]) / 2.0
            if "d" in a:
                r1 = r2 = _num(a["d"]) / 2.0
            center = False
            if "center" in a:
                if not isinstance(a["center"], bool):
                    raise ScadError("center must be a boolean")
                center = a["center"]
            base_y = 0.0 - h / 2.0 if center else 0.0
            # ConeFrustum::new(center, height, radius1, radius2, material): modules.rs:225-231
            return [sc.cone_frustum((0.0, base_y, 0.0), h, r1, r2, self.current_material())]
        if name == "quad":  # modules.rs:234-264
            a = self.convert_args(["q", "u", "v"], args)
            for key in ("q", "u", "v"):
                if key not in a:
                    raise ScadError(f"{key} is required")
            return [sc.quad(to_vector3(a["q"]), to_vector3(a["u"]), to_vector3(a["v"]), self.current_material())]
        if name == "translate":  # modules.rs:266-286
            a = self.convert_args(["v"], args)
            off = to_vector3(a["v"]) if "v" in a else (0.0, 0.0, 0.0)
            return [sc.translate(sc.group(nodes), off)]
        if name == "rotate":  # modules.rs:288-326
            a = self.convert_args(["a", "v"], args)
            if "a" not in a or not isinstance(a["a"], list):
                raise ScadError("rotate takes a vector of angles")
            ang = to_vector3(a["a"])
            node = sc.group(nodes)
            if ang[0] != 0.0:
                node = sc.rotate(node, (1.0, 0.0, 0.0), ang[0])
            if ang[1] != 0.0:
                node = sc.rotate(node, (0.0, 1.0, 0.0), ang[1])
            if ang[2] != 0.0:
                node = sc.rotate(node, (0.0, 0.0, 1.0), ang[2])
            return [node]
        if name == "scale":  # modules.rs:328-346
            a = self.convert_args(["v"], args)
            if "v" not in a:
                raise ScadError("scale: missing arg")
            v = to_vector3(a["v"])
            return [sc.scale(sc.group(nodes), v[0], v[1], v[2])]
        if name == "camera":
            self.make_camera(args)
            return []
        if name == "echo":  # modules.rs:443-473
            parts = []
            for nm, e in args:
                s = format_value(self.eval(e))
                parts.append(s if nm is None else f"{nm} = {s}")
            self.messages.append(("echo", ", ".join(parts)))
            return []
        raise ScadError(f'unknown identifier "{name}"')

    def for_loop(self, args, kids):  # mod.rs:247-305
        if len(args) != 1 or args[0][0] is None:
            raise ScadError("for takes one named argument")
        name, e = args[0]
        r = self.eval(e)
        if not isinstance(r, Range):
            raise ScadError("for takes a range")
        start, end = _num(r.start), _num(r.end)
        inc = 1.0 if r.increment is None else _num(r.increment)
        if (end >= start and inc <= 0.0) or (end < start and inc >= 0.0):
            raise ScadError("the increment has the wrong sign")
        i, out = start, []
        while not ((end >= start and i >= end) or (end < start and i <= end)):
            self.vars[-1][name] = i
            out += self.children(kids)
            i += inc
        return out

    def make_material(self, name, args):  # modules.rs:475-559
        if name == "color":
            a = self.convert_args(["c", "alpha"], args)
            if "alpha" in a:
                raise ScadError("color: alpha is not supported")
            if "c" not in a:
                raise ScadError("color: missing arg")
            return self._material(("lambertian",) + to_color(a["c"]))
        if name == "lambertian":
            a = self.convert_args(["c", "t"], args)
            if "c" in a:
                return self._material(("lambertian",) + to_color(a["c"]))
            if "t" in a and isinstance(a["t"], TextureRef):
                return self._material(("textured", a["t"].id))
            raise ScadError("lambertian: missing arg")
        if name == "dielectric":
            a = self.convert_args(["n"], args)
            if "n" not in a:
                raise ScadError("dielectric: missing arg")
            return self._material(("dielectric", _num(a["n"])))
        if name == "metal":
            a = self.convert_args(["c", "fuzz"], args)
            c = to_color(a["c"]) if "c" in a else (1.0, 1.0, 1.0)
            fuzz = _num(a["fuzz"]) if "fuzz" in a else 0.2
            return self._material(("metal",) + c + (fuzz,))
        a = self.convert_args(["c"], args)
        c = to_color(a["c"]) if "c" in a else (1.0, 1.0, 1.0)
        return self._material(("light",) + c)

    def make_camera(self, args):  # modules.rs:348-441
        names = ["image_width", "image_height", "samples_per_pixel", "max_depth", "vertical_fov", "look_from", "look_at", "up", "defocus_angle",
                 "focus_distance", "background", "aspect_ratio"]
        a = self.convert_args(names, args)
        cam = dict(aspect_ratio=1.0, image_width=100, samples_per_pixel=10, max_depth=10, background=(0.0, 0.0, 0.0), vertical_fov=90.0,
                   look_from=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), defocus_angle=0.0, focus_distance=10.0)
        seen_ar = seen_w = False
        if "aspect_ratio" in a:
            cam["aspect_ratio"] = _num(a["aspect_ratio"])
            seen_ar = True
        if "image_width" in a:
            cam["image_width"] = _rust_u32(_num(a["image_width"]))
            seen_w = True
        for key in ("samples_per_pixel", "max_depth"):
            if key in a:
                cam[key] = _rust_u32(_num(a[key]))
        for key in ("vertical_fov", "defocus_angle", "focus_distance"):
            if key in a:
                cam[key] = _num(a[key])
        if "image_height" in a:
            h = _num(a["image_height"])
            if seen_w:
                cam["aspect_ratio"] = float(cam["image_width"]) / h
            elif seen_ar:
                cam["image_width"] = _rust_u32(cam["aspect_ratio"] * h)
            else:
                cam["aspect_ratio"] = 1.0
                cam["image_width"] = _rust_u32(h)
        for key in ("look_from", "look_at", "up"):
            if key in a:
                cam[key] = to_vector3(a[key])
        if "background" in a:
            cam["background"] = to_color(a["background"])
        self.camera = cam

    # -- expressions (expr.rs)
    def lookup(self, name):
        for scope in reversed(self.vars):
            if name in scope:
                return scope[name]
        return None

    def eval(self, e):
        k = e[0]
        if k == "number":
            return e[1]
        if k == "string":
            return e[1]
        if k == "bool":
            return e[1]
        if k == "vector":
            return [self.eval(x) for x in e[1]]
        if k == "binary":
            return self.binary(e[1], self.eval(e[2]), self.eval(e[3]))
        if k == "neg":
            v = self.eval(e[1])
            if not _is_num(v):
                raise ScadError("unary minus takes a number")
            return -v
        if k == "not":
            return not _truthy(self.eval(e[1]))
        if k == "call":
            return self.call(e[1], e[2])
        if k == "range":
            return Range(self.eval(e[1]), self.eval(e[2]), None if e[3] is None else self.eval(e[3]))
        if k == "name":
            if e[1] == "PI":
                return math.pi
            if e[1] == "undef":
                return UNDEF
            v = self.lookup(e[1])
            if v is not None:
                return v
            if e[1] in self.functions:
                return FunctionRef(e[1])
            self.messages.append(("warning", f"Ignoring unknown variable '{e[1]}'"))
            return UNDEF
        if k == "index":
            lhs = self.eval(e[1])
            idx = _rust_i64(_num(self.eval(e[2])))
            if idx < 0:
                return UNDEF
            if isinstance(lhs, list):
                return lhs[idx] if idx < len(lhs) else UNDEF
            if isinstance(lhs, str):
                return lhs[idx] if idx < len(lhs) else UNDEF
            raise ScadError("cannot index this value")
        if k == "ternary":
            return self.eval(e[2]) if _truthy(self.eval(e[1])) else self.eval(e[3])
        if k == "field":
            lhs = self.eval(e[1])
            if isinstance(lhs, list) and e[2] in ("x", "y", "z"):
                return lhs["xyz".index(e[2])]
            raise ScadError(f"no field {e[2]}")
        raise ScadError(f"expression {k}")

    def binary(self, op, a, b):
        if op == "&&":
            return _truthy(a) and _truthy(b)
        if op == "||":
            return _truthy(a) or _truthy(b)
        if op in ("+", "-", "*", "/"):
            if _is_num(a):
                if _is_num(b):
                    return self._arith(op, a, b)
                if isinstance(b, list):
                    return [self.binary(op, a, x) for x in b]
                raise ScadError(f"unsupported operands for {op}")
            if isinstance(a, list):
                if _is_num(b):
                    return [self.binary(op, x, b) for x in a]
                if isinstance(b, list):
                    return [self.binary(op, x, y) for x, y in zip(a, b)]
            raise ScadError(f"unsupported operands for {op}")
        if op in ("^", "%"):
            if _is_num(a) and _is_num(b):
                if op == "^":
                    return _powf(a, b)
                try:
                    return math.fmod(a, b)  # f64 % f64
                except ValueError:
                    return float("nan")
            raise ScadError(f"unsupported operands for {op}")
        # comparisons (expr.rs:280-460)
        if _is_num(a):
            if not _is_num(b):
                return False
            return {"<": a < b, "<=": a <= b, ">": a > b, ">=": a >= b, "==": a == b, "!=": a != b}[op]
        if isinstance(a, list):
            if not isinstance(b, list):
                return False
            res = [self.binary(op, x, y) for x, y in zip(a, b)]
            if op == "==":
                return all(_truthy(x) for x in res)
            if op == "!=":
                return not all(not _truthy(x) for x in res)
            return res
        raise ScadError(f"unsupported operands for {op}")

    @staticmethod
    def _arith(op, a, b):
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if b == 0.0:
            if a == 0.0 or math.isnan(a):
                return float("nan")
            return math.copysign(float("inf"), a) * math.copysign(1.0, b)
        return a / b

    # -- functions (functions.rs)
    def call(self, name, args):
        f1 = {"abs": ("x", abs), "sign": ("x", lambda v: 0.0 if v == 0.0 else math.copysign(1.0, v)),
              "sin": ("degrees", lambda v: math.sin(math.radians(v))), "cos": ("degrees", lambda v: math.cos(math.radians(v))),
              "tan": ("degrees", lambda v: math.tan(math.radians(v))), "asin": ("x", lambda v: math.degrees(math.asin(v))),
              "acos": ("x", lambda v: math.degrees(math.acos(v))), "atan": ("x", lambda v: math.degrees(math.atan(v))),
              "floor": ("x", lambda v: float(math.floor(v))), "round": ("x", _round_half_away), "ceil": ("x", lambda v: float(math.ceil(v))),
              "ln": ("x", _ln), "log": ("x", _log10), "sqrt": ("x", _sqrt), "exp": ("x", math.exp)}
        if name in f1:
            arg, fn = f1[name]
            a = self.convert_args([arg], args)
            if arg not in a:
                raise ScadError(f"{name}: missing arg")
            try:
                return float(fn(_num(a[arg])))
            except (ValueError, OverflowError):
                return float("nan")
        if name in ("pow", "atan2"):
            names = ["base", "exponent"] if name == "pow" else ["y", "x"]
            a = self.convert_args(names, args)
            for nm in names:
                if nm not in a:
                    raise ScadError(f"missing {nm}")
            x, y = _num(a[names[0]]), _num(a[names[1]])
            return _powf(x, y) if name == "pow" else math.degrees(math.atan2(x, y))
        if name in ("min", "max"):
            vals = [self.eval(e) for _, e in args]
            better = (lambda p, q: p < q) if name == "min" else (lambda p, q: p > q)
            if not vals:
                return UNDEF
            if isinstance(vals[0], list):
                vals = vals[0]
                if not vals:
                    return UNDEF
            best = _num(vals[0])
            for v in vals:
                v = _num(v)
                if better(v, best):
                    best = v
            return best
        if name == "norm":
            a = self.convert_args(["v"], args)
            v = a.get("v", UNDEF)
            if not isinstance(v, list):
                return UNDEF
            return math.sqrt(sum(_powf(_num(x), 2.0) for x in v)) if v else 0.0
        if name == "cross":
            a = self.convert_args(["v1", "v2"], args)
            if "v1" not in a or "v2" not in a:
                raise ScadError("cross: missing argument")
            p, q = a["v1"], a["v2"]
            if not isinstance(p, list) or not isinstance(q, list):
                return UNDEF
            p, q = [_num(x) for x in p], [_num(x) for x in q]
            if len(p) != len(q):
                return UNDEF
            if len(p) == 2:
                return p[0] * q[1] - p[1] * q[0]
            if len(p) == 3:
                return [p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]]
            return UNDEF
        if name == "concat":
            out = []
            for _, e in args:
                v = self.eval(e)
                out += v if isinstance(v, list) else [v]
            return out
        if name == "lookup":
            return self.fn_lookup(args)
        if name == "rands":
            return self.fn_rands(args)
        if name.startswith("is_"):
            a = self.convert_args(["x"], args)
            if "x" not in a:
                raise ScadError(f"{name}: missing arg")
            v = a["x"]
            tests = {"is_undef": v is UNDEF, "is_bool": isinstance(v, bool), "is_num": _is_num(v), "is_string": isinstance(v, str),
                     "is_list": isinstance(v, list), "is_function": isinstance(v, FunctionRef)}
            if name in tests:
                return tests[name]
        if name == "checker":  # functions.rs:465-487
            a = self.convert_args(["scale", "even", "odd"], args)
            scale = _num(a["scale"]) if "scale" in a else 1.0
            even = to_color(a["even"]) if "even" in a else (0.0, 0.0, 0.0)
            odd = to_color(a["odd"]) if "odd" in a else (1.0, 1.0, 1.0)
            return TextureRef(self.sc.tex_checker(scale, self.sc.tex_solid(*even), self.sc.tex_solid(*odd)))
        if name == "perlin_turbulence":  # functions.rs:489-511
            a = self.convert_args(["scale", "turbulence_depth"], args)
            scale = _num(a["scale"]) if "scale" in a else 1.0
            depth = _rust_u32(_num(a["turbulence_depth"])) if "turbulence_depth" in a else 1
            return TextureRef(self.sc.tex_perlin_turbulence(scale, depth, _scenes.perlin_tables(self.scene_rng)))
        if name == "image":  # functions.rs:513-532
            a = self.convert_args(["filename"], args)
            if "filename" not in a or not isinstance(a["filename"], str):
                raise ScadError("image: filename required")
            if self.image_loader is None:
                raise ScadError(f'failed to get image "{a["filename"]}": no image loader')
            return TextureRef(self.sc.tex_image(np.asarray(self.image_loader(a["filename"]), dtype=np.uint8)))
        # user function (functions.rs:583-607)
        if name not in self.functions:
            raise ScadError(f"missing function {name}")
        params, body = self.functions[name]
        a = self.convert_args(params, args)
        self.vars.append(dict(a))
        try:
            return self.eval(body)
        finally:
            self.vars.pop()

    def fn_lookup(self, args):  # functions.rs:75-149
        a = self.convert_args(["key", "table"], args)
        if "key" not in a or "table" not in a:
            raise ScadError("lookup: missing argument")
        key = _num(a["key"])
        if not isinstance(a["table"], list):
            raise ScadError("table must be a vector")
        rows = []
        for row in a["table"]:
            if not isinstance(row, list):
                raise ScadError("table must be a list of lists")
            if len(row) != 2:
                raise ScadError("table row must be list of 2 elements")
            rows.append((_num(row[0]), _num(row[1])))
        if not rows:
            raise ScadError("table must have at least 1 row")
        if key <= rows[0][0]:
            return rows[0][1]
        if key >= rows[-1][0]:
            return rows[-1][1]
        last = rows[0]
        for row in rows:
            if key == row[0]:
                return row[1]
            if key <= row[0]:
                p = (key - last[0]) / (row[0] - last[0])
                return last[1] + p * (row[1] - last[1])
            last = row
        raise ScadError("lookup: unreachable")

    def fn_rands(self, args):  # functions.rs:534-581
        a = self.convert_args(["min_value", "max_value", "value_count", "seed_value"], args)
        for nm in ("min_value", "max_value", "value_count"):
            if nm not in a:
                raise ScadError(f"{nm} required")
        lo, hi = _num(a["min_value"]), _num(a["max_value"])
        count = _num(a["value_count"])
        count = 0 if math.isnan(count) else int(min(max(count, 0.0), 1.8446744073709552e19))
        if "seed_value" in a:
            raise ScadError("rands with a seed is not supported")
        if hi < lo:
            lo, hi = hi, lo
        out = []
        for _ in range(count):
            normalized = float(self.rng.next_u64()) / 18446744073709551615.0  # u64 -> f64, then / (u64::MAX as f64)
            out.append(lo + normalized * (hi - lo))
        return out


def _powf(a, b):
    try:
        return math.pow(a, b)
    except (ValueError, OverflowError):
        if a < 0 and not float(b).is_integer():
            return float("nan")
        return float("inf")


def _ln(v):
    if v == 0.0:
        return float("-inf")
    if v < 0.0 or math.isnan(v):
        return float("nan")
    return math.log(v)


def _log10(v):
    if v == 0.0:
        return float("-inf")
    if v < 0.0 or math.isnan(v):
        return float("nan")
    return math.log10(v)


def _sqrt(v):
    return float("nan") if v < 0.0 or math.isnan(v) else math.sqrt(v)


def _round_half_away(v):  # f64::round
    if math.isnan(v) or math.isinf(v):
        return v
    return math.copysign(math.floor(abs(v) + 0.5), v)


def run_openscad(source: str, backend, scene_seed=1, image_loader=None):
    """crates/openscad/src/lib.rs:88-128: tokenize, parse, interpret.  Returns (Scene, messages); messages are
    (level, text) pairs with level in {"error", "warning", "echo"}.  The scene is committed."""
    statements, msgs = parse(source)
    it = Interpreter(backend.new_scene(), scene_seed=scene_seed, image_loader=image_loader)
    sc, more = it.run(statements)
    return sc, msgs + more


def echo_output(source: str, backend) -> str:
    """The messages' texts, one per line: what the reference's interpreter tests compare (interpreter/tests.rs:25-33)."""
    _, msgs = run_openscad(source, backend)
    return "".join(text + "\n" for _, text in msgs)
