"""Import the upstream reference (read-only, /root/reference) in THIS container only.

The reference needs four pure-plumbing packages that are not installed here
(tasklogger, pygsp, future, deprecated).  None of them touches arithmetic, so
they are replaced by in-memory stubs.  Nothing from the
reference is copied into the repository: this module only puts it on sys.path.

Used by tools/make_golden.py (fixture generation) and by nothing that runs on
the GPU box.
"""
import contextlib
import sys
import types

REFERENCE_ROOT = "/root/reference"


def import_reference():
    if "graphtools" in sys.modules and getattr(
        sys.modules["graphtools"], "__file__", ""
    ).startswith(REFERENCE_ROOT):
        return sys.modules["graphtools"]

    tl = types.ModuleType("tasklogger")

    class _Logger:
        name = "graphtools"

        def set_level(self, *a, **k):
            pass

        def log_info(self, *a, **k):
            pass

        def log_debug(self, *a, **k):
            pass

        def log_warning(self, *a, **k):
            pass

        @contextlib.contextmanager
        def log_task(self, *a, **k):
            yield

    tl.get_tasklogger = lambda name="x": _Logger()
    sys.modules["tasklogger"] = tl

    pg, pgg, pgu = (types.ModuleType(n) for n in ("pygsp", "pygsp.graphs", "pygsp.utils"))

    class _G:
        def __init__(self, W=None, **kw):
            self.W = W

    pgg.Graph = _G
    pg.graphs = pgg
    pg.utils = pgu
    sys.modules.update({"pygsp": pg, "pygsp.graphs": pgg, "pygsp.utils": pgu})

    fu, fuu = types.ModuleType("future"), types.ModuleType("future.utils")

    def with_metaclass(meta, *bases):
        class metaclass(type):
            def __new__(cls, name, this_bases, d):
                return meta(name, bases, d)

        return type.__new__(metaclass, "temporary_class", (), {})

    fuu.with_metaclass = with_metaclass
    fu.utils = fuu
    sys.modules.update({"future": fu, "future.utils": fuu})

    dp = types.ModuleType("deprecated")
    dp.deprecated = lambda *a, **k: (lambda f: f)
    sys.modules["deprecated"] = dp

    sys.path.insert(0, REFERENCE_ROOT)
    import graphtools  # noqa: E402

    assert graphtools.graphs.NUMBA_AVAILABLE is False
    return graphtools
