"""Module-level knobs the reference keeps in pyradUtilities (ut:804-805, ut:841).

``BASE_RESOLUTION`` is looked up dynamically at every use, as the reference does
(cls:41, 287, 416, 567, 660-662, 672, 704, ...): change it with ``set_resolution_multiplier``
or by assigning ``settings.BASE_RESOLUTION`` before building layers.
"""
RES_MULTIPLIER = 1
BASE_RESOLUTION = .01 * RES_MULTIPLIER
VERSION = '1.75'
DEVICE = 0          # HIP device index used by the default engine
PINNED_POOL_BYTES = 32 << 20     # page-locked host staging allocated with the engine (0: allocate on first use)


def set_resolution_multiplier(mult):
    global RES_MULTIPLIER, BASE_RESOLUTION
    RES_MULTIPLIER = mult
    BASE_RESOLUTION = .01 * mult


ACCURACY = "exact"  # "exact": the device reproduces the reference's fp64 values to 1e-14 (default);
                    # "budget": <= 1e-9 relative on the absorption coefficient, still fp64, faster (lbl_set_option "accuracy")


def set_accuracy(mode):
    """Accuracy mode of the default engine, now and for engines created later."""
    global ACCURACY
    if mode not in ("exact", "budget"):
        raise ValueError("accuracy mode must be 'exact' or 'budget'")
    ACCURACY = mode
    from . import engine
    if engine._engine is not None and engine._engine.ctx.h is not None:
        engine._engine.ctx.set_option("accuracy", 1 if mode == "budget" else 0)


LAYER_STEP = "merged"   # how a Layer's absorption coefficient / transmittance are computed when its line lists are due:
                        # "merged": ONE accumulate job over the layer's merged, factor-weighted line lists, the absorption
                        #   coefficient sum_m f_m sum_iso xs_iso (cls:707-712) accumulated directly (lbl_layer_merged_step_dev;
                        #   Atmosphere.transmission: lbl_layers_merged_accumulate_dev + lbl_column_fold_dev).  Isotope / Molecule
                        #   cross sections are then produced when somebody asks for them (the reference's own lazy protocol,
                        #   progressCrossSection, cls:32-88);
                        # "per-list": one job and one cross-section array per line list, then the sweep over them
                        #   (lbl_layer_step_dev / lbl_column_step_dev)


def set_layer_step(mode):
    global LAYER_STEP
    if mode not in ("merged", "per-list"):
        raise ValueError("layer step must be 'merged' or 'per-list'")
    LAYER_STEP = mode
