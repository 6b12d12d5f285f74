"""Scoring of registration results: the metric of Registration/registration_dataset/evaluate_rt.py.

* RTE = norm of the translation of P_pred^-1 * P_gt; RRE = sum of the absolute xyz Euler angles (degrees) of
  its rotation (evaluate_rt.py:21-29); success iff RTE < 2 m and RRE < 5 deg (evaluate_rt.py:16-18).
* Result files are the CSV of main.py:220-222: ``idx1,idx2,t_x,t_y,t_z,q_w,q_x,q_y,q_z`` with the quaternion
  stored w-first (evaluate_rt.py:65-74); the header may or may not carry np.savetxt's leading "# ".
* ``evaluate_rt`` keeps the reference's accounting: row 0 is the header, the success RATE is divided by the
  number of lines INCLUDING the header (evaluate_rt.py:103), the averages by the number of successes.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

__all__ = ["get_P_from_Rt", "get_P_diff", "is_registration_successful", "read_reg_results", "reg_result_row_to_array",
           "evaluate_rt", "pose_from_row"]

RTE_MAX_M = 2.0
RRE_MAX_DEG = 5.0


def get_P_from_Rt(R, t):
    P = np.eye(4)
    P[:3, :3] = np.asarray(R, dtype=np.float64)
    P[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    return P


def get_P_diff(P_pred_np, P_gt_np):
    """-> (RTE metres, RRE degrees)."""
    delta = np.linalg.inv(np.asarray(P_pred_np, dtype=np.float64)) @ np.asarray(P_gt_np, dtype=np.float64)
    rte = float(np.linalg.norm(delta[:3, 3]))
    euler = Rotation.from_matrix(delta[:3, :3]).as_euler("xyz", degrees=True)
    return rte, float(np.abs(euler).sum())


def is_registration_successful(P_pred_np, P_gt_np):
    rte, rre = get_P_diff(P_pred_np, P_gt_np)
    return (rte < RTE_MAX_M and rre < RRE_MAX_DEG), rte, rre


def read_reg_results(file_path, splitter=","):
    """List of rows, each a list of stripped strings; the header line is row 0 (evaluate_rt.py:53-62)."""
    with open(file_path, "r") as f:
        return [[item.strip() for item in line.split(splitter)] for line in f if line]


def reg_result_row_to_array(reg_result_row):
    """-> (idx1, idx2, t (3,), scipy Rotation); the file stores q as w,x,y,z (evaluate_rt.py:65-74)."""
    idx1, idx2 = int(reg_result_row[0]), int(reg_result_row[1])
    t = np.array([float(v) for v in reg_result_row[2:5]])
    qw, qx, qy, qz = (float(v) for v in reg_result_row[5:9])
    return idx1, idx2, t, Rotation.from_quat([qx, qy, qz, qw])


def pose_from_row(row):
    idx1, idx2, t, rot = reg_result_row_to_array(row)
    return idx1, idx2, get_P_from_Rt(rot.as_matrix(), t)


def evaluate_rt(gt_file_path, predict_file_path, verbose=False):
    """-> (success rate, mean RTE of the successes, mean RRE of the successes), evaluate_rt.py:77-112."""
    gt = read_reg_results(gt_file_path)
    pred = read_reg_results(predict_file_path)
    if len(gt) != len(pred):
        raise AssertionError("ground truth and prediction files differ in length")
    ok, rte_sum, rre_sum = 0, 0.0, 0.0
    for g_row, p_row in zip(gt[1:], pred[1:]):
        g1, g2, Pg = pose_from_row(g_row)
        p1, p2, Pp = pose_from_row(p_row)
        if (g1, g2) != (p1, p2):
            raise AssertionError(f"pair mismatch: {(g1, g2)} vs {(p1, p2)}")
        success, rte, rre = is_registration_successful(Pp, Pg)
        if success:
            ok += 1
            rte_sum += rte
            rre_sum += rre
            if verbose:
                print(p_row)
    rate = ok / len(gt)
    avg_rte = rte_sum / ok  # the reference divides by zero when nothing succeeds; so do we (ZeroDivisionError)
    avg_rre = rre_sum / ok
    print("Registration successful rate: %.2f, successful counter: %d, \n"
          "average Relative Translation Error (RTE): %.2f, average Relative Rotation Error (RRE): %.2f" % (rate, ok, avg_rte, avg_rre))
    return rate, avg_rte, avg_rre
