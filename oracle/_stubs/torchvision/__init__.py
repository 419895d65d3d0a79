"""Stub of `torchvision` for importing the reference's YOLOX modules in this image (torchvision is not installed).
TEST INFRASTRUCTURE: only the two NMS entry points models/detection/yolox/utils/boxes.py:57-66 looks up are declared;
they are never called by the tests (post-processing is outside SURVEY.md §8)."""


class _Ops:
    @staticmethod
    def nms(boxes, scores, iou_threshold):
        raise NotImplementedError('torchvision stub: nms is outside the hot-path scope')

    @staticmethod
    def batched_nms(boxes, scores, idxs, iou_threshold):
        raise NotImplementedError('torchvision stub: batched_nms is outside the hot-path scope')


ops = _Ops()
