"""Module-level knobs the reference keeps in pyradUtilities (ut:804-805, ut:841).

``BASE_RESOLUTION`` is looked up dynamically at every use, as the reference does
(cls:41, 287, 416, 567, 660-662, 672, 704, ...): change it with ``set_resolution_multiplier``
or by assigning ``settings.BASE_RESOLUTION`` before building layers.
"""
RES_MULTIPLIER = 1
BASE_RESOLUTION = .01 * RES_MULTIPLIER
VERSION = '1.75'
DEVICE = 0          # HIP device index used by the default engine


def set_resolution_multiplier(mult):
    global RES_MULTIPLIER, BASE_RESOLUTION
    RES_MULTIPLIER = mult
    BASE_RESOLUTION = .01 * mult
