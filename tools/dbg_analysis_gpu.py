import sys, numpy as np
sys.path.insert(0,'/root/repo')
from tests import ctu_trace as T
from tests.test_analysis_cpu import fixture_picture, GOLDEN
from x265_amd import capi
L = capi.load(); capi.check(L.xmi_init(0))
g = np.load(GOLDEN)
for rep in range(3):
    pic, before, rec_md5 = fixture_picture(g, "416x240_q32_p1")
    r = T.Runner(L, "xmi_analysis_", pic["w"], pic["h"])
    r.set_state(before, pic)
    out = r.run(pic)
    st = r.get_state(pic["rows"], pic["cols"])
    pic["rec"] = (out["rec_y"], out["rec_u"], out["rec_v"])
    msgs = T.compare(pic, out, st)
    print("rep", rep, msgs[:20])
    d = np.argwhere(out["coef_y"] != pic["coef_y"]); print(" coef_y diffs", len(d), d[:5].tolist())
    d = np.argwhere(out["mv"] != pic["mv"]); print(" mv diffs", len(d), d[:5].tolist())
    d = np.argwhere(out["rd_cost"] != pic["rd_cost"]); print(" rd diffs", len(d), d[:8].tolist(), out["rd_cost"].ravel()[:8], pic["rd_cost"].ravel()[:8])
    r.close()
