#!/usr/bin/env python
"""Golden report files: runs the COMPILED REFERENCE binary (oracle/_ref/SOAPnuke) on seeded
synthetic FASTQ for every case in tests/report_util.py and stores its report .txt files (data
only) under tests/golden/reports/<case>/.   python tests/golden/make_golden_reports.py"""
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import report_util as R  # noqa: E402
import snk_testlib as T  # noqa: E402

assert os.path.exists(T.REF_BIN), "make -C oracle ref"
for case in R.REPORT_CASES:
    d, p = R.case_inputs(case)
    work = tempfile.mkdtemp(prefix="snkgold_")
    ref = R.run_reference_cli(case, d, work)
    dst = os.path.join(HERE, "reports", case[0])
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    for f in (R.REPORT_FILES_PE if case[1] else R.REPORT_FILES_SE):
        shutil.copy(os.path.join(ref, f), os.path.join(dst, f))
    shutil.rmtree(work)
    print(case[0], "ok")
