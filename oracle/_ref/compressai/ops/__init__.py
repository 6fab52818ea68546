def compute_padding(in_h, in_w, *, out_h=None, out_w=None, min_div=1):
    if out_h is None: out_h = (in_h + min_div - 1) // min_div * min_div
    if out_w is None: out_w = (in_w + min_div - 1) // min_div * min_div
    left = (out_w - in_w) // 2; right = out_w - in_w - left
    top = (out_h - in_h) // 2; bottom = out_h - in_h - top
    return (left, right, top, bottom), (-left, -right, -top, -bottom)
