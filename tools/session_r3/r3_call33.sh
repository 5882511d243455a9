tools/bin/probe_conv12 8 576
tools/bin/probe_conv12 32 576 | tail -8
