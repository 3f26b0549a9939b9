"""Feature tables and histories (upstream KLT 1.3.4 storeFeatures.c; the reference only has the empty
KLT_FeatureTable / KLT_FeatureHistory stubs at klt.py:272-283).  Host-side Python; the records use the 16-byte layout of
`klt_feat`, so a table row can go to / come from a device feature buffer without conversion."""
from __future__ import print_function

from .error import KLTError
from .klt import KLT_FeatureHistory, KLT_FeatureTable, new_feature_list, shared_store
from .selectGoodFeatures import features_to_array


def KLTCreateFeatureList(nFeatures):
    return new_feature_list(nFeatures)


def KLTCreateFeatureHistory(nFrames):
    return KLT_FeatureHistory(nFrames)


def KLTCreateFeatureTable(nFrames, nFeatures):
    return KLT_FeatureTable(nFrames, nFeatures)


def _check_frame(ft, frame, who):
    if frame < 0 or frame >= ft.nFrames:
        KLTError("({0}) Given frame number ({1}) is not in range of feature table (which has {2} frames)".format(
            who, frame, ft.nFrames))


def _check_feature(ft, feat, who):
    if feat < 0 or feat >= ft.nFeatures:
        KLTError("({0}) Given feature number ({1}) is not in range of feature table (which has {2} features)".format(
            who, feat, ft.nFeatures))


def KLTStoreFeatureList(fl, ft, frame):
    """Copy a feature list into row `frame` of the table."""
    _check_frame(ft, frame, "KLTStoreFeatureList")
    if len(fl) != ft.nFeatures:
        KLTError("(KLTStoreFeatureList) FeatureList and FeatureTable must have the same number of features")
    ft.rec[frame] = features_to_array(fl)


def KLTExtractFeatureList(fl, ft, frame):
    """Copy row `frame` of the table into an existing feature list."""
    _check_frame(ft, frame, "KLTExtractFeatureList")
    if len(fl) != ft.nFeatures:
        KLTError("(KLTExtractFeatureList) FeatureList and FeatureTable must have the same number of features")
    row = ft.rec[frame]
    store = shared_store(fl)
    if store is not None:
        store.x[:], store.y[:], store.val[:] = row["x"], row["y"], row["val"]
        store.xint[:] = False
        store.yint[:] = False
        store.changed()
        return
    for feat, x, y, v in zip(fl, row["x"].tolist(), row["y"].tolist(), row["val"].tolist()):
        feat.x, feat.y, feat.val = x, y, v


def KLTStoreFeatureHistory(fh, ft, feat):
    _check_feature(ft, feat, "KLTStoreFeatureHistory")
    if fh.nFrames != ft.nFrames:
        KLTError("(KLTStoreFeatureHistory) FeatureHistory and FeatureTable must have the same number of frames")
    ft.rec[:, feat] = fh.rec


def KLTExtractFeatureHistory(fh, ft, feat):
    _check_feature(ft, feat, "KLTExtractFeatureHistory")
    if fh.nFrames != ft.nFrames:
        KLTError("(KLTExtractFeatureHistory) FeatureHistory and FeatureTable must have the same number of frames")
    fh.rec[:] = ft.rec[:, feat]
