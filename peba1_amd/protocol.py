"""Protocol P_1 as the reference's driver runs it (src/main.cpp:533-586): the client's
encrypted sample is matched against the stored encrypted template with Function_f, the
server draws two random bytes r0, r1, Function_g turns the encrypted match bit into an
encryption of one of them, the client decrypts it and is authenticated when it returns r1.

The driver reproduces the reference's behaviour, not its intent: Function_f's bit is 1 when
the distance EXCEEDS the bound and Function_g returns r1 for bit 1, so the reference
"authenticates" exactly the samples that do not match, and for bit 0 its
|1 - 0| = 255 (bootsSUBNbit, Math.cpp:137-138) makes y = 255*r0 mod 256 rather than r0.
`function_f=circuits.function_f_fast` swaps in the optimised DAG (same match bit).

Host plumbing only; every gate goes through the boots* C ABI of libtfhe-hip.so.  Each
function runs in deferred mode and is flushed by the decryption (or the explicit flush)
that follows it.

    python -m peba1_amd.protocol --nslots 128            # genuine and impostor run, one GPU
"""
import argparse
import json
import time

from . import api, circuits

MAX_BITSIZE = 24        # 3 * bitsize: width of the distance and of result_b (main.cpp:46)


def run_p1(params, key, sample, template, bound_match, r0, r1, bitsize=8, cloud=None, function_f=None):
    """One protocol run.  `key` is the client's secret keyset (encrypts, decrypts); `cloud`
    (default: its embedded cloud keyset) is all the evaluating side uses.  Returns a dict with
    the decrypted y, whether the client was authenticated (y == r1, main.cpp:578) and timings."""
    ev = key if cloud is None else cloud
    t0 = time.perf_counter()
    enc_template = circuits.EncryptedVector(params, template, bitsize, key).to_device()
    enc_sample = circuits.EncryptedVector(params, sample, bitsize, key).to_device()
    enc_bound = circuits.encrypt_number(params, bound_match, MAX_BITSIZE, key)
    enc_r0 = circuits.encrypt_number(params, r0, bitsize, key)           # main.cpp:553-559
    enc_r1 = circuits.encrypt_number(params, r1, bitsize, key)
    t_enc = time.perf_counter()
    was_deferred = api.get_deferred()
    api.set_deferred(True)
    try:
        enc_b = api.CiphertextArray(params, MAX_BITSIZE)
        (function_f or circuits.function_f)(enc_b, enc_sample, enc_template, enc_bound, bitsize, ev)    # main.cpp:538
        levels_f = api.flush()
        t_f = time.perf_counter()
        enc_y = api.CiphertextArray(params, bitsize + 1)
        circuits.function_g(enc_y, enc_b, enc_r0, enc_r1, bitsize, ev)                 # main.cpp:564
        levels_g = api.flush()
        t_g = time.perf_counter()
    finally:
        api.set_deferred(was_deferred)
    y = circuits.decrypt_number(enc_y, key, bitsize)                                   # main.cpp:571-575
    b = int(enc_b.decrypt(key)[0])
    return {"y": y, "r0": r0, "r1": r1, "match_bit": b, "authenticated": y == r1,
            "seconds": {"encrypt": t_enc - t0, "function_f": t_f - t_enc, "function_g": t_g - t_f},
            "levels": {"function_f": levels_f, "function_g": levels_g}}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--nslots", type=int, default=128)
    ap.add_argument("--bound", type=int, default=256)
    ap.add_argument("--seed", type=lambda v: int(v, 0), default=0x5EBA1)
    ap.add_argument("--fast", action="store_true", help="Function_f through the optimised DAG (circuits_fast.cpp)")
    a = ap.parse_args()
    params = api.ParameterSet(128)
    key = api.SecretKeySet(params, a.seed + 1)
    template = [(37 * i + 11) % 255 for i in range(a.nslots)]             # SURVEY 8c inputs
    runs = {"genuine": [t + 1 for t in template], "impostor": [(91 * i + 5) % 256 for i in range(a.nslots)]}
    for name, sample in runs.items():
        out = run_p1(params, key, sample, template, a.bound, r0=17, r1=99,
                     function_f=circuits.function_f_fast if a.fast else None)
        out["distance"] = sum((x - y) ** 2 for x, y in zip(sample, template))
        print(json.dumps({"run": name, **out}))
    key.close()


if __name__ == "__main__":
    main()
