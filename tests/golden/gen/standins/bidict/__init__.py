"""Minimal stand-in for the third-party `bidict` package (absent from this image, no network).

TEST INFRASTRUCTURE ONLY - used by tests/golden/gen/make_goldens.py so that the *unmodified* reference at
/root/reference imports here.  It is a pure container: a dict that also exposes `.inverse` (value -> key).
No codec arithmetic lives in it.  Never imported by the product package.
"""


class bidict(dict):
    @property
    def inverse(self):
        return {v: k for k, v in self.items()}
