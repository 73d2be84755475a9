"""x265_amd: MI355X-native motion-estimation engine behind the x265 ME / primitives API."""
