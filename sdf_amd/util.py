"""angle helpers re-exported by ``from sdf import *`` (reference sdf/util.py:1-6)"""
import math

pi = math.pi

degrees = math.degrees
radians = math.radians
