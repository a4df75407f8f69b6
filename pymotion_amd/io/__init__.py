"""Motion-capture file ingest feeding the GPU hot path (``bvh``)."""
