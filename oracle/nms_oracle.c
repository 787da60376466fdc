/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of greedy box NMS with torchvision's public CPU semantics, which the reference
 * calls as torch.ops.torchvision.nms (celldetection/ops/cpn.py:181,211,216,223;
 * celldetection_scripts/cpn_inference.py:407,426).  torchvision is a third-party dependency that is NOT
 * under /root/reference and is unpinned there (requirements.txt: bare "torchvision"); the algorithm
 * restated here is its published one:
 *   areas = (x2-x1)*(y2-y1); visit boxes in stable descending-score order (`order`, computed by the caller);
 *   a visited, unsuppressed box i is kept and suppresses every later j with
 *   inter/(area_i+area_j-inter) > thr, inter = max(0,xx2-xx1)*max(0,yy2-yy1); NaN compares false.
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC nms_oracle.c -o libnms_oracle.so  (done by cpn_oracle.py)
 */
#include <stdlib.h>

long nms_oracle(const float *boxes, const long *order, long n, float thr, long *keep) {
    unsigned char *suppressed = (unsigned char *) calloc((size_t) n, 1);
    long nk = 0;
    for (long a = 0; a < n; ++a) {
        long i = order[a];
        if (suppressed[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float iarea = (ix2 - ix1) * (iy2 - iy1);
        for (long c = a + 1; c < n; ++c) {
            long j = order[c];
            if (suppressed[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1 > 0.f ? xx2 - xx1 : 0.f;
            float h = yy2 - yy1 > 0.f ? yy2 - yy1 : 0.f;
            float inter = w * h;
            float jarea = (boxes[4 * j + 2] - boxes[4 * j]) * (boxes[4 * j + 3] - boxes[4 * j + 1]);
            float ovr = inter / (iarea + jarea - inter);
            if (ovr > thr) suppressed[j] = 1;
        }
    }
    free(suppressed);
    return nk;
}
