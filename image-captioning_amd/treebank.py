"""nltk.word_tokenize for the caption pipeline (the reference: `from nltk.tokenize import word_tokenize`,
dense_img_cap_separate_models/preprocess.py:5,52,61,72; dense_img_cap/preprocess.py likewise).

NLTK is a third-party dependency the reference neither vendors nor pins (no requirements file; its API use fits nltk 3.2-3.4)
and it is not installable here.  word_tokenize there is  sent_tokenize (Punkt, a trained model)  followed by the Penn Treebank
word tokenizer: Robert MacIntyre's published sed script as a fixed list of regular-expression substitutions, plus the unicode
quote rules nltk adds in word_tokenize.  The substitution list below restates that published algorithm rule for rule (same
order, which matters); tests/test_host_logic.py holds it to the known answers of NLTK's own documentation.  The sentence
splitter is an approximation of Punkt (whose trained parameter file is not available): a sentence ends at a token-final
'.', '?' or '!' followed by whitespace unless the token is one of a short list of abbreviations -- for Punkt, too, a
period-final token that is not a known abbreviation ends the sentence whatever the case of the next word, which is what
matters for the lower-cased region phrases this path feeds it.
"""
import re

# --- Treebank rules, in application order -------------------------------------------------------------------------
_STARTING_QUOTES = [
    (re.compile(u'([«“‘])', re.U), r' \1 '),          # word_tokenize's extra opening-quote rule
    (re.compile(r'^\"'), r'``'),
    (re.compile(r'(``)'), r' \1 '),
    (re.compile(r'([ (\[{<])"'), r'\1 `` '),
]
_PUNCTUATION = [
    (re.compile(u'([^\\.])(\\.)([\\]\\)}>"\'»”’ ]*)\\s*$', re.U), r'\1 \2 \3 '),   # word_tokenize's final-period rule
    (re.compile(r'([:,])([^\d])'), r' \1 \2'),
    (re.compile(r'([:,])$'), r' \1 '),
    (re.compile(r'\.\.\.'), r' ... '),
    (re.compile(r'[;@#$%&]'), r' \g<0> '),
    (re.compile(r'([^\.])(\.)([\]\)}>"\']*)\s*$'), r'\1 \2\3 '),
    (re.compile(r'[?!]'), r' \g<0> '),
    (re.compile(r"([^'])' "), r"\1 ' "),
]
_PARENS_BRACKETS = (re.compile(r'[\]\[\(\)\{\}\<\>]'), r' \g<0> ')
_DOUBLE_DASHES = (re.compile(r'--'), r' -- ')
_ENDING_QUOTES = [
    (re.compile(u'([»”’])', re.U), r' \1 '),          # word_tokenize's extra closing-quote rule
    (re.compile(r'"'), " '' "),
    (re.compile(r'(\S)(\'\')'), r'\1 \2 '),
    (re.compile(r"([^' ])('[sS]|'[mM]|'[dD]|') "), r"\1 \2 "),
    (re.compile(r"([^' ])('ll|'LL|'re|'RE|'ve|'VE|n't|N'T) "), r"\1 \2 "),
]
_CONTRACTIONS2 = [re.compile(p) for p in (r"(?i)\b(can)(not)\b", r"(?i)\b(d)('ye)\b", r"(?i)\b(gim)(me)\b", r"(?i)\b(gon)(na)\b",
                                          r"(?i)\b(got)(ta)\b", r"(?i)\b(lem)(me)\b", r"(?i)\b(mor)('n)\b", r"(?i)\b(wan)(na)\s")]
_CONTRACTIONS3 = [re.compile(p) for p in (r"(?i) ('t)(is)\b", r"(?i) ('t)(was)\b")]


def treebank_tokenize(text, word_tokenize_rules=True):
    """TreebankWordTokenizer().tokenize(text); with word_tokenize_rules the three unicode-quote / final-period rules
    nltk.word_tokenize prepends are active as well."""
    skip = 0 if word_tokenize_rules else 1
    for rx, sub in _STARTING_QUOTES[skip:]:
        text = rx.sub(sub, text)
    for rx, sub in _PUNCTUATION[skip:]:
        text = rx.sub(sub, text)
    text = _PARENS_BRACKETS[0].sub(_PARENS_BRACKETS[1], text)
    text = _DOUBLE_DASHES[0].sub(_DOUBLE_DASHES[1], text)
    text = " " + text + " "
    for rx, sub in _ENDING_QUOTES[skip:]:
        text = rx.sub(sub, text)
    for rx in _CONTRACTIONS2:
        text = rx.sub(r' \1 \2 ', text)
    for rx in _CONTRACTIONS3:
        text = rx.sub(r' \1 \2 ', text)
    return text.split()


# Punkt's English model carries a learned list of abbreviation types (period-final tokens that do NOT end a sentence).  The pickle is
# not available here; this list restates the common ones (titles, company forms, months, U.S. state forms of its Wall Street Journal
# training text, clock / measure forms).  A period-final token outside the list ends the sentence, as in Punkt's first pass.
_ABBREVIATIONS = frozenset((
    "mr mrs ms messrs dr prof gen col lt maj sgt capt adm cmdr rep reps sen sens gov rev hon st vs etc jr sr inc co cos corp ltd bros "
    "no nos mt ft approx dept univ ave blvd rd e.g i.e u.s u.k u.n a.m p.m ph.d m.d b.a m.a d.c n.y n.j n.h n.c s.c n.m w.va "
    "jan feb mar apr jun jul aug sep sept oct nov dec mon tue tues wed thu thur thurs fri sat sun "
    "calif mass conn fla pa va ga ill mich minn tenn wash wis colo ariz ala okla ore kan ky la nev neb mo miss ind del tex").split())
_SENT_END = re.compile(r'(\S+?[.?!]+["\'\)\]]*)(\s+)')


def sent_tokenize(text):
    """Approximate Punkt: split after a token ending in . ? ! (+ closing quotes / brackets) that is followed by
    whitespace, unless the token (without the final period) is a listed abbreviation or a single letter (an initial)."""
    out, start = [], 0
    for m in _SENT_END.finditer(text):
        tok = m.group(1).rstrip('"\')]')
        core = tok.rstrip('.?!').lower()
        if tok.endswith('.') and (core in _ABBREVIATIONS or (len(core) == 1 and core.isalpha())):
            continue
        out.append(text[start:m.end(1)])
        start = m.end()
    rest = text[start:].strip()
    if rest:
        out.append(rest)
    return [s.strip() for s in out if s.strip()]


def word_tokenize(text):
    """nltk.word_tokenize(text): sentences, then Treebank tokens of each."""
    return [tok for sent in sent_tokenize(text) for tok in treebank_tokenize(sent)]
