"""ORACLE -- test infrastructure only (see oracle/ops.py).  CPU restatement of the reference's
kernel_type='special' path; never imported by pyopenvino_amd."""
