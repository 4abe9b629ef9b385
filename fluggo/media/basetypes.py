"""Value types the process module hands out and accepts: v2i, v2f, box2i, box2f, rgba.

`fluggo.media.process` imports this module at init and builds these types when it returns windows and
colours (the reference does the same: src/process/basetypes.c:117-148).  They are plain tuples with
names, so anything that accepts the reference's types accepts these:
  v2i(x, y) / v2f(x, y)                        -- also accept one 2-tuple
  box2i(min, max) / box2i(x0, y0, x1, y1)      -- inclusive bounds; empty when max < min on an axis
  rgba(r, g, b, a=1.0)
YAML tags (!v2i, !box2i, !v2f, !box2f, !rgba, !rational) are registered when PyYAML is importable.
"""
import collections
import fractions


def _pair(cls_name, conv):
    base = collections.namedtuple("_" + cls_name, "x y")

    class vec(base):
        __slots__ = ()

        def __new__(cls, x=0, y=0):
            if isinstance(x, tuple):
                x, y = x
            return base.__new__(cls, conv(x), conv(y))

        def __add__(self, other):
            return type(self)(self[0] + other[0], self[1] + other[1])

        def __sub__(self, other):
            return type(self)(self[0] - other[0], self[1] - other[1])

        def __repr__(self):
            return "%s(%r, %r)" % (cls_name, self.x, self.y)

    vec.__name__ = vec.__qualname__ = cls_name
    return vec


v2i = _pair("v2i", int)
v2f = _pair("v2f", float)


def _box(cls_name, vec, one):
    base = collections.namedtuple("_" + cls_name, "min max")

    class box(base):
        __slots__ = ()

        def __new__(cls, min=None, max=None, max_x=None, max_y=None):
            if max_x is not None and max_y is not None:      # box(x0, y0, x1, y1)
                min, max = vec(min, max), vec(max_x, max_y)
            elif isinstance(min, base):                        # copy
                min, max = min
            elif min is None:                                  # canonical empty box
                min, max = vec(0, 0), vec(-1, -1)
            return base.__new__(cls, vec(min), vec(max))

        @property
        def width(self):
            return _max0(self.max.x - self.min.x + one)

        @property
        def height(self):
            return _max0(self.max.y - self.min.y + one)

        def size(self):
            if self.empty():
                return vec()
            return self.max - self.min + vec(one, one)

        def empty(self):
            return not bool(self)

        def __bool__(self):
            return self.max.x >= self.min.x and self.max.y >= self.min.y

        def __repr__(self):
            return "%s(%r, %r)" % (cls_name, self.min, self.max)

    box.__name__ = box.__qualname__ = cls_name
    return box


def _max0(v):
    return v if v > 0 else type(v)(0)


box2i = _box("box2i", v2i, 1)
box2f = _box("box2f", v2f, 1.0)

_rgba = collections.namedtuple("_rgba", "r g b a")


class rgba(_rgba):
    __slots__ = ()

    def __new__(cls, r=0.0, g=0.0, b=0.0, a=1.0):
        return _rgba.__new__(cls, float(r), float(g), float(b), float(a))

    def __repr__(self):
        return "rgba({0.r:.6}, {0.g:.6}, {0.b:.6}, {0.a:.6})".format(self)


def _register_yaml():
    try:
        import yaml
    except ImportError:
        return
    yaml.add_representer(fractions.Fraction, lambda d, v: d.represent_sequence("!rational", [v.numerator, v.denominator]))
    yaml.add_constructor("!rational", lambda l, n: fractions.Fraction(*l.construct_sequence(n)))
    for tag, cls in (("!v2i", v2i), ("!v2f", v2f)):
        yaml.add_representer(cls, lambda d, v, tag=tag: d.represent_sequence(tag, [v.x, v.y]))
        yaml.add_constructor(tag, lambda l, n, cls=cls: cls(*l.construct_sequence(n)))
    for tag, cls in (("!box2i", box2i), ("!box2f", box2f)):
        yaml.add_representer(cls, lambda d, v, tag=tag: d.represent_sequence(tag, [list(v.min), list(v.max)]))
        yaml.add_constructor(tag, lambda l, n, cls=cls: cls(*[tuple(p) for p in l.construct_sequence(n, deep=True)]))
    yaml.add_representer(rgba, lambda d, v: d.represent_sequence("!rgba", list(v)))
    yaml.add_constructor("!rgba", lambda l, n: rgba(*l.construct_sequence(n)))


_register_yaml()
