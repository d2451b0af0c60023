for v in 0; do echo "variant $v"; KSGPU_CSR_VARIANT=$v python scripts/csr_probe.py 216 csr 2>&1 | grep -v "^$"; done
