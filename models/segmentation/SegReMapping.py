"""Drop-in for the reference's models/segmentation/SegReMapping.py (numpy class, :5-76): post-processing of
segmentation label maps before masked cWCT.  Same results; one histogram + one lookup-table pass per call
instead of a `seg == label` scan per label.  `mapping_name` is the reference's ade20k_semantic_rel.npy
([150,150] int: column l lists the labels most related to l, best first), which is data of the reference and is
passed in by path (image_transfer.py:35 --label_mapping), or an array."""
import numpy as np


class SegReMapping:
    def __init__(self, mapping_name, min_ratio=0.01):
        self.label_mapping = np.load(mapping_name) if isinstance(mapping_name, (str, bytes)) else np.asarray(mapping_name)
        self.min_ratio = min_ratio
        self.label_ipt = []

    def _lut_apply(self, seg, labels, new_labels):
        lut = np.arange(max(int(seg.max()) + 1, int(max(new_labels, default=0)) + 1), dtype=seg.dtype)
        lut[np.asarray(labels, dtype=np.int64)] = np.asarray(new_labels, dtype=seg.dtype)
        return lut[seg]

    def cross_remapping(self, content_seg, style_seg):
        """:19-48 — content labels absent from the style map move to the most related label present in the style."""
        content_seg = np.asarray(content_seg)
        cont = [int(l) for l in np.unique(content_seg)]
        style = set(int(l) for l in np.unique(style_seg))
        new = list(cont)
        for idx, s in enumerate(cont):
            if s in style or s in self.label_ipt:
                continue
            for j in range(self.label_mapping.shape[0]):
                cand = int(self.label_mapping[j, s])
                if cand in style:
                    new[idx] = cand
                    break
        return self._lut_apply(content_seg, cont, new)

    def self_remapping(self, seg):
        """:51-76 — labels covering less than min_ratio of the image move to the most related label that covers
        at least min_ratio (ratios are those of the ORIGINAL map, as in the reference)."""
        seg = np.asarray(seg)
        n_pixels = seg.shape[0] * seg.shape[1]
        counts = np.bincount(seg.reshape(-1).astype(np.int64))
        labels = [int(l) for l in np.nonzero(counts)[0]]
        ratio = {l: np.float32(counts[l]) / n_pixels for l in labels}
        new = list(labels)
        for i, cur in enumerate(labels):
            if ratio[cur] < self.min_ratio:
                for j in range(self.label_mapping.shape[0]):
                    cand = int(self.label_mapping[j, cur])
                    if cand in ratio and ratio[cand] >= self.min_ratio:
                        new[i] = cand
                        break
        return self._lut_apply(seg, labels, new)
