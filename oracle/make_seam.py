#!/usr/bin/env python3
"""oracle/make_seam.py SRC DST -- write DST = the reference's src/operator-run.c with ONE statement inserted at the top of
qnnp_run_operator (in front of the `switch (op->ukernel_type)` of src/operator-run.c:646): operators that live on the
device go to the gfx950 build, everything else falls through to the reference's own dispatch. DST is a build product
(oracle/_ref/hybrid/, git-ignored); nothing of the reference is copied into the repository. TEST INFRASTRUCTURE."""
import sys

src, dst = sys.argv[1], sys.argv[2]
text = open(src).read()
head = "enum qnnp_status qnnp_run_operator(qnnp_operator_t op, pthreadpool_t threadpool)\n{\n"
assert text.count(head) == 1, "the reference's qnnp_run_operator was not found where it is expected"
seam = ("extern int qnnp_hybrid_owns(qnnp_operator_t op);\n"
        "extern enum qnnp_status qnnp_gfx950_run_operator(qnnp_operator_t op, pthreadpool_t threadpool);\n" + head +
        "  if (qnnp_hybrid_owns(op)) return qnnp_gfx950_run_operator(op, threadpool);   /* the seam: INTEGRATION.md section 2 */\n")
open(dst, "w").write(text.replace(head, seam))
