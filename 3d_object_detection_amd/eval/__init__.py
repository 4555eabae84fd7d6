"""eval.* of the reference (eval/iou.py, eval/eval.py): rotated overlaps on the device, AP bookkeeping on the host.
`install()` of the parent package also exposes these modules as `eval.iou` / `eval.eval`."""
__all__ = ["iou", "eval"]
