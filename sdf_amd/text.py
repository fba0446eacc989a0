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


def _load_image(thing):
    """reference sdf/text.py:11-16"""
    from PIL import Image
    if isinstance(thing, str):
        return Image.open(thing)
    elif isinstance(thing, (np.ndarray, np.generic)):
        return Image.fromarray(thing)
    return Image.fromarray(np.array(thing))


def _fit(aspect, width, height):
    if width is None and height is None:
        height = 1
    if width is None:
        width = height * aspect
    if height is None:
        height = width / aspect
    return (width, height)


def measure_text(name, text, width=None, height=None):
    """reference sdf/text.py:18-28"""
    from PIL import ImageFont
    font = ImageFont.truetype(name, 96)
    x0, y0, x1, y1 = font.getbbox(text)
    return _fit((x1 - x0) / (y1 - y0), width, height)


def measure_image(thing, width=None, height=None):
    """reference sdf/text.py:30-40"""
    im = _load_image(thing)
    w, h = im.size
    return _fit(w / h, width, height)


@d2.sdf2
def text(font_name, text, width=None, height=None, pixels=PIXELS, points=512):
    """reference sdf/text.py:42-63"""
    from PIL import Image, ImageFont, ImageDraw
    font = ImageFont.truetype(font_name, points)
    p = 0.2
    x0, y0, x1, y1 = font.getbbox(text)
    px = int((x1 - x0) * p)
    py = int((y1 - y0) * p)
    tw = x1 - x0 + 1 + px * 2
    th = y1 - y0 + 1 + py * 2
    im = Image.new('L', (tw, th))
    draw = ImageDraw.Draw(im)
    draw.text((px - x0, py - y0), text, font=font, fill=255)
    return _sdf(width, height, pixels, px, py, im)


@d2.sdf2
def image(thing, width=None, height=None, pixels=PIXELS):
    """reference sdf/text.py:65-68"""
    im = _load_image(thing).convert('L')
    return _sdf(width, height, pixels, 0, 0, im)


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
