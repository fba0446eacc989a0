"""Text / image leaves (reference sdf/text.py:1-153): a glyph string or a picture becomes a 2-D
signed-distance TEXTURE on the host (PIL rasterisation + scipy's exact Euclidean distance
transform, like the reference), and a `texture2d` leaf that the tape interpreter samples
bilinearly on the device (csrc/sdf_interp.h L_TEXTURE2D), with the reference's fallback rectangle
outside the texture.  Same names, arguments and defaults as the reference; the result is an SDF2,
so `.extrude(...)` etc. work as usual.

The host preprocessing needs Pillow and scipy (the reference needs them too); they are imported
lazily so that the rest of the package works without them.
"""
import numpy as np

from . import d2
from .ir import Node, unwrap

PIXELS = 2 ** 22


def _as_pil(source):
    """a PIL image from a file name, an array or anything `np.array` accepts (reference sdf/text.py:11-16)"""
    from PIL import Image
    if isinstance(source, str):
        return Image.open(source)
    pixels = source if isinstance(source, (np.ndarray, np.generic)) else np.array(source)
    return Image.fromarray(pixels)


_load_image = _as_pil          # (the reference's name for it)


def _fit(aspect, width, height):
    """the reference's sizing rule: a missing side follows from the other and the aspect ratio; neither given: height 1"""
    if width is None and height is None:
        height = 1
    if width is None:
        width = height * aspect
    if height is None:
        height = width / aspect
    return (width, height)


def _glyph_box(font_file, string, size):
    """(font, left, top, columns, rows) of `string` set in `font_file` at `size` points"""
    from PIL import ImageFont
    font = ImageFont.truetype(font_file, size)
    left, top, right, bottom = font.getbbox(string)
    return font, left, top, right - left, bottom - top


def measure_text(name, text, width=None, height=None):
    """reference sdf/text.py:18-28"""
    _, _, _, cols, rows = _glyph_box(name, text, 96)
    return _fit(cols / rows, width, height)


def measure_image(thing, width=None, height=None):
    """reference sdf/text.py:30-40"""
    cols, rows = _as_pil(thing).size
    return _fit(cols / rows, width, height)


MARGIN = 0.2        # of the glyph box, on every side (reference sdf/text.py:48)


@d2.sdf2
def text(font_name, text, width=None, height=None, pixels=PIXELS, points=512):
    """reference sdf/text.py:42-63: the string rendered white on black into an 8-bit canvas one pixel larger than its glyph box
    plus the margin on every side, then the distance texture of that canvas"""
    from PIL import Image, ImageDraw
    font, left, top, cols, rows = _glyph_box(font_name, text, points)
    pad = (int(cols * MARGIN), int(rows * MARGIN))
    canvas = Image.new('L', (cols + 1 + 2 * pad[0], rows + 1 + 2 * pad[1]))
    ImageDraw.Draw(canvas).text((pad[0] - left, pad[1] - top), text, font=font, fill=255)
    return _sdf(width, height, pixels, pad[0], pad[1], canvas)


@d2.sdf2
def image(thing, width=None, height=None, pixels=PIXELS):
    """reference sdf/text.py:65-68: any picture, as 8-bit grey, without padding"""
    return _sdf(width, height, pixels, 0, 0, _as_pil(thing).convert('L'))


def distance_texture(mask):
    """signed distance (pixels) of a boolean mask: negative inside, positive outside
    (reference sdf/text.py:81-87)"""
    import scipy.ndimage as nd
    a = np.asarray(mask, dtype=bool)
    inside = -nd.distance_transform_edt(a)
    outside = nd.distance_transform_edt(~a)
    texture = np.zeros(a.shape)
    texture[a] = inside[a]
    texture[~a] = outside[~a]
    return texture


def _sdf(width, height, pixels, px, py, im):
    """reference sdf/text.py:70-136 up to the closure; the closure itself (`f`, :116-134, with
    `_bilinear_interpolate`, :138-153) is the `texture2d` leaf evaluated on the device"""
    tw, th = im.size
    factor = (pixels / (tw * th)) ** 0.5
    if factor < 1:
        tw, th = int(round(tw * factor)), int(round(th * factor))
        px, py = int(round(px * factor)), int(round(py * factor))
        im = im.resize((tw, th))
    im = im.convert('1')
    texture = distance_texture(np.array(im))
    return _texture_node(texture, width, height, px, py)


def _texture_node(texture, width=None, height=None, px=0, py=0):
    texture = np.array(texture, dtype=np.float64)          # (th, tw), copied: it is scaled below
    th, tw = texture.shape
    pw = tw - px * 2
    ph = th - py * 2
    width, height = _fit(pw / ph, width, height)
    x0 = -width / 2
    y0 = -height / 2
    x1 = width / 2
    y1 = height / 2
    scale = width / tw
    texture *= scale
    rect = unwrap(d2.rectangle((width / 2, height / 2)))    # fallback outside the texture (:112)
    return Node('texture2d', [x0, y0, x1, y1, pw, ph, px, py, tw, th, *rect.params], meta={'blob': texture})


@d2.sdf2
def texture_sdf(texture, width=None, height=None, px=0, py=0):
    """the leaf for a ready-made distance texture in PIXEL units (rows = image rows, top row first):
    world size (width, height) as in `image`, `px`/`py` = padding pixels around the content
    (reference sdf/text.py:98-134)"""
    return _texture_node(texture, width, height, px, py)
