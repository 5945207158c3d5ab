"""The `3d_pairs` record schema of the reference's result JSON (exps/stage3_root2/test_util.py:134-158).

Two writers exist in the reference and both are kept:
  * `frame_record` -- save_result (:146-158): one record per FRAME with every registered person
    (`run_inference`: gt lists empty; `generate_result`: gt_3d = annotation columns 4.., gt_2d = columns 0..3);
  * `train_records` -- save_result_for_train_refine (:134-143): one record per matched PERSON (root score != 0),
    the input format of RefineNet training (dataset/p2p_dataset.py:11-36)."""
import numpy as np


def _lists(a):
    return np.asarray(a).tolist()


def _arrays(a):
    return np.array(a)                 # a private copy: the pipeline's pinned result buffers are reused


def frame_record(pred_2d, pred_3d, root_d, image_path, gt_bodys=None, as_lists=True):
    """as_lists=False keeps the numbers as numpy arrays (copies): the inference loop then does no per-element Python
    work; `to_jsonable` turns such records into the reference's nested lists when the file is written."""
    cv = _lists if as_lists else _arrays
    rec = {"pred_2d": cv(pred_2d), "pred_3d": cv(pred_3d), "root_d": cv(root_d), "image_path": image_path}
    if gt_bodys is not None:
        rec["gt_3d"] = cv(gt_bodys[:, :, 4:])
        rec["gt_2d"] = cv(gt_bodys[:, :, :4])
    else:
        rec["gt_3d"], rec["gt_2d"] = [], []
    return rec


def train_records(pred_2d, pred_3d, root_d, gt_bodys, root_n=2, as_lists=True):
    cv = _lists if as_lists else _arrays
    out = []
    for i, body in enumerate(pred_3d):
        if body[root_n][3] != 0:
            out.append({"pred_3d": cv(body), "pred_2d": cv(pred_2d[i]), "gt_3d": cv(gt_bodys[i][:, 4:7]),
                        "root_d": float(root_d[i])})
    return out


def to_jsonable(records):
    """Records built with as_lists=False -> the nested-list form json.dump writes (same numbers: ndarray.tolist())."""
    return [{k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in r.items()} for r in records]


def kept_annotations(annotation, root_idx=2):
    """test.py:76-80: annotations whose root joint is visible (score > 1); zero padding rows drop out too."""
    annotation = np.asarray(annotation)
    return annotation[annotation[:, root_idx, 3] > 1]


def annotation_camera(gt_bodys, scale):
    """[9] float64 scale,img_w,img_h,net_w,net_h,f_x,f_y,cx,cy from the first kept annotation (test.py:84-93):
    11-column annotations carry (f_x, f_y, cx, cy) in columns 7..10; shorter ones only f in column 7, and the
    principal point defaults to the image centre."""
    g = lambda k: float(np.asarray(scale[k]).reshape(-1)[0])
    cam = [g("scale"), g("img_width"), g("img_height"), g("net_width"), g("net_height")]
    if gt_bodys.shape[2] < 11:
        cam += [float(gt_bodys[0, 0, 7]), float(gt_bodys[0, 0, 7]), g("img_width") / 2, g("img_height") / 2]
    else:
        cam += [float(gt_bodys[0, 0, 7]), float(gt_bodys[0, 0, 8]), float(gt_bodys[0, 0, 9]), float(gt_bodys[0, 0, 10])]
    return np.asarray(cam, np.float64)
