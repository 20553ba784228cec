"""Import shim for the `semantic_version` package (absent here); only what
kpal/__init__.py:40-41,101-104 touches.  Used only by tools/gen_golden.py."""


class Version(object):
    def __init__(self, s):
        self._s = str(s)
        self.tuple = tuple(int(x) for x in self._s.split('-')[0].split('+')[0].split('.')[:3])

    def __str__(self):
        return self._s


class SimpleSpec(object):
    def __init__(self, spec):
        self.spec = spec

    def __contains__(self, v):
        return (1, 0, 0) <= v.tuple < (2, 0, 0)


Spec = SimpleSpec
