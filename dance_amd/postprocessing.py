"""The reference's `detector_postprocess` for the contour heads, with the masks made on the GPU.

    detector_postprocess(results, output_height, output_width)
        <- core/modeling/dsnake_baseline/postprocessing.py:24-76 (FcosSnake): scale boxes and
           contours to the output resolution, clip, drop empty boxes, pred_masks = RLEs
    dance_detector_postprocess(..., results, output_height, output_width, ..., re_comp_box)
        <- core/modeling/postprocessing.py:287-358 (Dance): the same without clipping the
           contours and without masks (the reference's evaluator makes them later,
           coco_evaluation.py:352-357 -- use dance_amd.rle.get_polygon_rles there)

`results` is a detectron2 `Instances` (or anything with `.image_size`, `.has()`,
`.get_fields()`, boolean indexing and a `(image_size, **fields)` constructor); boxes need
detectron2 `Boxes`' `scale / clip / nonempty`.  Host glue only: the arithmetic of the masks is
`dance_amd.rle` (csrc/raster.hip); the contours never leave the device.
"""
from .rle import get_polygon_rles

__all__ = ["get_polygon_rles", "detector_postprocess", "dance_detector_postprocess"]


def _rescale(results, output_height, output_width):
    scale_x = output_width / results.image_size[1]
    scale_y = output_height / results.image_size[0]
    results = type(results)((output_height, output_width), **results.get_fields())
    if results.has("pred_boxes"):
        output_boxes = results.pred_boxes
    elif results.has("proposal_boxes"):
        output_boxes = results.proposal_boxes
    else:
        raise ValueError("No boxes in instance prediction!")
    output_boxes.scale(scale_x, scale_y)
    output_boxes.clip(results.image_size)
    return results[output_boxes.nonempty()], scale_x, scale_y


def _rescale_path(results, scale_x, scale_y, clip):
    if results.has("pred_path"):
        path = results.pred_path                       # (N, iterations, V, 2), in place as the reference
        path[..., 0] *= scale_x
        path[..., 1] *= scale_y
        if clip:
            h, w = results.image_size
            path[..., 0].clamp_(min=0, max=w)
            path[..., 1].clamp_(min=0, max=h)


def detector_postprocess(results, output_height, output_width):
    """dsnake_baseline/postprocessing.py:24-76."""
    results, scale_x, scale_y = _rescale(results, output_height, output_width)
    if not results.has("pred_polys"):
        raise ValueError("No pred_polys in instance prediction!")
    _rescale_path(results, scale_x, scale_y, clip=True)
    results.pred_polys.scale(scale_x, scale_y)
    results.pred_polys.clip(results.image_size)
    results.pred_masks = get_polygon_rles(results.pred_polys.flatten(), (output_height, output_width))
    return results


def dance_detector_postprocess(semantic_filter, semantic_filter_th, mask_result_src, results, output_height,
                               output_width, roi_size, need_concave_hull, re_comp_box):
    """core/modeling/postprocessing.py:287-358 (the live branch: results with pred_polys)."""
    results, scale_x, scale_y = _rescale(results, output_height, output_width)
    if not results.has("pred_polys"):
        return None                                    # the reference falls off its `if` here
    _rescale_path(results, scale_x, scale_y, clip=True)
    if results.has("ext_points"):
        results.ext_points.scale(scale_x, scale_y)
    results.pred_polys.scale(scale_x, scale_y)
    if re_comp_box:
        boxes = results.pred_polys.get_box()
        results.pred_boxes = type(results.pred_boxes)(boxes)
    return results
