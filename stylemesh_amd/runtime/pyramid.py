"""Style-image pyramid sizes (reference model/losses/content_and_style_losses.py:83-133, ``image_pyramid`` with
``reverse=True``): level l halves the image l times (``int(x / 2**l)``) but never drops below a minimum side of
256 (the first too-small level is replaced by a resize whose SHORT side is exactly 256); the list is reversed
up to that entry (smallest first) and padded with the original size."""


def image_pyramid_sizes(h: int, w: int, levels, minimum_size: int = 256):
    sizes, min_entry, min_index = [], None, len(levels)
    for i, level in enumerate(levels):
        if level == 0:
            sizes.append((h, w))
            continue
        hd, wd = int(h / 2 ** level), int(w / 2 ** level)
        if hd < minimum_size or wd < minimum_size:
            if min_entry is None:
                if w > h:
                    min_entry = (minimum_size, int(w * minimum_size / h))
                else:
                    min_entry = (int(h * minimum_size / w), minimum_size)
                min_index = i
            sizes.append(min_entry)
        else:
            sizes.append((hd, wd))
    out = sizes[:min_index + 1][::-1]
    while len(out) < len(sizes):
        out.append((h, w))
    return out
