"""Text / image leaves (reference sdf/text.py:42-153).  Out of scope of the hot path
(host-side PIL + EDT preprocessing feeding a texture leaf); the names
exist so that ``from sdf import *`` keeps exporting them."""


def _todo(name):
    def f(*args, **kwargs):
        raise NotImplementedError(
            'sdf_amd does not implement %s yet (sampled 2-D texture leaves are a "next" row, '
            'DESIGN.md)' % name)
    f.__name__ = name
    return f


measure_image = _todo('measure_image')
measure_text = _todo('measure_text')
image = _todo('image')
text = _todo('text')
